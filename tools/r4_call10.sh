#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4j; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -q -m gpu -x --timeout 600 -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
grep -n "passed\|failed\|FAILED\|fault\|^E  " $O/pytest.log | cut -c1-300 | head -30
