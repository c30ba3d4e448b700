#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4n; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 600 python tools/d2h_bench.py 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_multi.py -q -m gpu --timeout 600 -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log
