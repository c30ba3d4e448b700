#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s; mkdir -p $O
V="rh3.bin eu_rc8.bin eu_ras7.bin eu_rw6.bin"
( echo "== C3 full"; timeout 150 python tools/ab_fast.py --rounds 2 --frames 60 $V
  echo "== C4 full"; timeout 100 python tools/ab_fast.py --workload triangles-10m-8k --rounds 1 --frames 40 $V
) > $O/ab5.txt 2>&1
grep -v "^---- " $O/ab5.txt | cut -c1-260
