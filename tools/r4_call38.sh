#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s; mkdir -p $O
timeout 400 python -m pytest tests -q -x -m gpu --timeout 200 -p no:cacheprovider -k "parity or styling or stress or fuzz or contract" > $O/pytest9.log 2>&1
rc=$?; tail -2 $O/pytest9.log
[ $rc -ne 0 ] && exit 1
V="fin.bin gp1.bin"
( echo "== C3 full"; timeout 150 python tools/ab_fast.py --rounds 3 --frames 60 $V
  echo "== circles"; timeout 100 python tools/ab_fast.py --workload circles-20k --rounds 1 --frames 60 $V
) > $O/ab11.txt 2>&1
grep -v "^---- " $O/ab11.txt | grep "crc\|==\|identical" | cut -c1-260
