#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s; mkdir -p $O
V="head.bin rh2.bin rh2.bin@FORMA_HIP_DEBUG=no_ras_hist"
bash tools/ab_prof_all.sh $V head.bin rh2.bin
bash tools/ab_prof_all.sh triangles-10m-8k head.bin rh2.bin
( echo "== C3 full"; timeout 400 python tools/ab_fast.py --rounds 2 --frames 60 $V
  echo "== C4 full"; timeout 400 python tools/ab_fast.py --workload triangles-10m-8k --rounds 1 --frames 40 $V
  echo "== C3 band F=3"; AB_BAND=59,76 AB_INFLIGHT=3 timeout 400 python tools/ab_fast.py --rounds 1 --frames 150 $V
) > $O/ab2.txt 2>&1
grep -v "^---- " $O/ab2.txt | grep -v "crc" | cut -c1-260
