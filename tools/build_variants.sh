#!/bin/bash
# tools/build_variants.sh name1:"-DFLAG=..." name2:"..." -> forma_amd/csrc/variants/name.bin (a full libforma_hip.so each)
# for same-box A/B runs with tools/ab_bench.sh
set -e
cd forma_amd/csrc
mkdir -p variants
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -fno-gpu-flush-denormals-to-zero -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result"
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  [ "$name" = "$spec" ] && flags=""
  rm -f *.o
  make -s -j8 HIPFLAGS="$BASE $flags" >/dev/null
  cp libforma_hip.so variants/$name.bin
  echo "built $name [$flags]"
done
rm -f *.o
make -s -j8 >/dev/null
