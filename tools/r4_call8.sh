#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4h; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest "tests/test_gpu_round2.py::test_value_range_digits_on_a_power_of_two_canvas" -q -m gpu --timeout 600 -p no:cacheprovider 2>&1 | grep -E "passed|failed|^E  |^tests.*Error|assert" | cut -c1-240 | head -30
