#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s; mkdir -p $O
V="fin.bin occ5.bin occ7.bin occ8.bin"
( echo "== C3 full"; timeout 200 python tools/ab_fast.py --rounds 2 --frames 60 $V
  echo "== circles"; timeout 150 python tools/ab_fast.py --workload circles-20k --rounds 1 --frames 60 $V
) > $O/ab12.txt 2>&1
grep -v "^---- " $O/ab12.txt | grep "crc\|==\|identical" | cut -c1-260
