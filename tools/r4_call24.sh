#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s; mkdir -p $O
V="rh2.bin rh3.bin"
bash tools/ab_prof_all.sh $V $V
( echo "== C3 full"; timeout 400 python tools/ab_fast.py --rounds 2 --frames 60 $V
  echo "== C4 full"; timeout 400 python tools/ab_fast.py --workload triangles-10m-8k --rounds 1 --frames 40 $V
  echo "== C3 band F=3"; AB_BAND=59,76 AB_INFLIGHT=3 timeout 400 python tools/ab_fast.py --rounds 2 --frames 150 $V
) > $O/ab3.txt 2>&1
grep -v "^---- " $O/ab3.txt | grep -v "crc" | cut -c1-260
timeout 900 python -m pytest tests -q -m gpu -x --timeout 600 -p no:cacheprovider -k "not stress" > $O/pytest3.log 2>&1
tail -3 $O/pytest3.log
