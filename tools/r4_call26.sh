#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s; mkdir -p $O
timeout 240 python -m pytest tests/test_gpu_round2.py -q -x --timeout 100 -p no:cacheprovider -k "rasterizer or value_range or frames_in_flight" > $O/pytest4.log 2>&1
rc=$?; tail -3 $O/pytest4.log
[ $rc -ne 0 ] && exit 1
V="rh3.bin rh4.bin"
( echo "== C3 full"; timeout 100 python tools/ab_fast.py --rounds 2 --frames 60 $V
  echo "== C4 full"; timeout 100 python tools/ab_fast.py --workload triangles-10m-8k --rounds 1 --frames 40 $V
  echo "== cubics"; timeout 100 python tools/ab_fast.py --workload cubics-1080p --rounds 1 --frames 60 $V
) > $O/ab4.txt 2>&1
grep -v "^---- " $O/ab4.txt | cut -c1-260
