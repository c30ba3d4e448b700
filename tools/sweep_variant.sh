#!/bin/bash
# tools/sweep_variant.sh name.bin ...: tools/sort_sweep.py under rocprofv3 for each library variant (same box)
cd forma_amd/csrc; cp libforma_hip.so /tmp/lib_keep.so; cd ../..
export TMPDIR=/tmp
for v in "$@"; do
  cp forma_amd/csrc/variants/$v forma_amd/csrc/libforma_hip.so
  rm -rf /tmp/sw_$v
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/sw_$v -- python $OLDPWD/tools/sort_sweep.py run > /dev/null 2>&1)
  echo "== $v"; python tools/sort_sweep.py parse /tmp/sw_$v
done
cp /tmp/lib_keep.so forma_amd/csrc/libforma_hip.so
