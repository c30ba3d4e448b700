#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4l; mkdir -p $O
export PYTHONUNBUFFERED=1
# background load: the painters of another process keep the GPU busy while the poison test loops
( for i in 1 2 3 4 5 6 7 8; do timeout 120 python -m pytest tests/test_gpu_cache.py -q -m gpu -p no:cacheprovider > /dev/null 2>&1; done ) &
BG=$!
fails=0
for i in $(seq 1 24); do
  timeout 300 python -m pytest "tests/test_gpu_stress.py::test_nothing_reads_what_it_did_not_write" -q -m gpu --timeout 300 -p no:cacheprovider > $O/p$i.log 2>&1 || { fails=$((fails+1)); grep -n "fault\|Assert" $O/p$i.log | head -5 | cut -c1-300; }
done
echo "poison loop: $fails failures of 24"
kill $BG 2>/dev/null; wait $BG 2>/dev/null
