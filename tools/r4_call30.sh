#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s; mkdir -p $O
timeout 200 python -m pytest tests/test_gpu_round2.py -q -x --timeout 100 -p no:cacheprovider -k "sort or value_range" > $O/pytest6.log 2>&1
rc=$?; tail -2 $O/pytest6.log
[ $rc -ne 0 ] && exit 1
V="rh3.bin k9.bin"
( echo "== C4 full"; timeout 100 python tools/ab_fast.py --workload triangles-10m-8k --rounds 2 --frames 40 $V
  echo "== C3 full"; timeout 150 python tools/ab_fast.py --rounds 1 --frames 60 $V
) > $O/ab8.txt 2>&1
grep -v "^---- " $O/ab8.txt | grep -v crc | cut -c1-260
timeout 900 python -m pytest tests -q -m gpu -x --timeout 600 -p no:cacheprovider > $O/pytest7.log 2>&1
tail -3 $O/pytest7.log
