#!/bin/bash
# F=4: which commit changed it?  the band proxy against older builds of the library, 4 and 8 hardware queues
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04q; mkdir -p $O
python tools/ab_fast.py --rounds 0 > /dev/null 2>&1
V=$PWD/forma_amd/csrc/variants
for lib in v35d5369.bin v43718c7.bin ""; do
for q in 4 8; do
  if [ -n "$lib" ]; then export FORMA_HIP_LIB=$V/$lib; else unset FORMA_HIP_LIB; fi
  GPU_MAX_HW_QUEUES=$q timeout 300 python tools/band_proxy.py --slots 3,4 --frames 300 > $O/tmp.log 2>&1
  echo "lib=$lib q=$q"; grep '"band"' $O/tmp.log | sed 's/.*us_per_frame/us_per_frame/'
done; done
