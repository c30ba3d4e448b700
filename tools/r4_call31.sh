#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s; mkdir -p $O
V="k9.bin k9.bin@FORMA_HIP_DEBUG=carry_slices=1 k9.bin@FORMA_HIP_DEBUG=carry_slices=2 k9.bin@FORMA_HIP_DEBUG=carry_slices=3 k9.bin@FORMA_HIP_DEBUG=carry_slices=4 k9.bin@FORMA_HIP_DEBUG=no_small_carry"
( echo "== C3 full"; timeout 200 python tools/ab_fast.py --rounds 1 --frames 60 $V
  echo "== C4 full"; timeout 200 python tools/ab_fast.py --workload triangles-10m-8k --rounds 1 --frames 40 $V
  echo "== C3 band F=3"; AB_BAND=59,76 AB_INFLIGHT=3 timeout 200 python tools/ab_fast.py --rounds 1 --frames 150 $V
) > $O/ab9.txt 2>&1
grep -v "^---- " $O/ab9.txt | grep "crc\|==" | cut -c1-260
