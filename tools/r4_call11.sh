#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4k; mkdir -p $O
export PYTHONUNBUFFERED=1
for i in 1 2 3; do
timeout 900 python -m pytest tests -q -m gpu -n 6 --timeout 600 -p no:cacheprovider > $O/pytest$i.log 2>&1
echo "run $i rc=$?"; tail -3 $O/pytest$i.log | cut -c1-200
done
for i in 1 2 3 4 5 6; do timeout 300 python -m pytest "tests/test_gpu_stress.py" -q -m gpu -k "poison" --timeout 300 -p no:cacheprovider 2>&1 | tail -1; done
