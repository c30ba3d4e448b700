#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s; mkdir -p $O
timeout 120 env FORMA_HIP_LIB=$PWD/forma_amd/csrc/variants/kpt14.bin python -m pytest tests/test_gpu_round2.py -q -x --timeout 100 -p no:cacheprovider -k "sort" > $O/pytest5.log 2>&1
rc=$?; tail -2 $O/pytest5.log
[ $rc -ne 0 ] && exit 1
V="rh3.bin kpt14.bin kpt12.bin"
( echo "== C3 full"; timeout 150 python tools/ab_fast.py --rounds 2 --frames 60 $V
  echo "== C4 full"; timeout 100 python tools/ab_fast.py --workload triangles-10m-8k --rounds 1 --frames 40 $V
  echo "== cubics"; timeout 100 python tools/ab_fast.py --workload cubics-1080p --rounds 1 --frames 60 $V
  echo "== circles"; timeout 100 python tools/ab_fast.py --workload circles-20k --rounds 1 --frames 60 $V
) > $O/ab7.txt 2>&1
grep -v "^---- " $O/ab7.txt | grep -v crc | cut -c1-260
