#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4m; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -q -m gpu -n 6 --timeout 600 -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log
timeout 600 python tools/d2h_bench.py 2>&1 | tail -4
V="base.bin new4.bin"
( echo "== C3 full"; timeout 400 python tools/ab_fast.py --rounds 1 --frames 60 $V ) 2>&1 | grep -v "^---- \|identical" | cut -c1-250
