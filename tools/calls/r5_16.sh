#!/bin/bash
# call 16: strips and the heavy-first order only with one frame in flight — round-4 library against round 5's, pipelined and per call
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r5_16; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
export AB_KERNELS=1
timeout 400 python tools/ab_fast.py --rounds 3 --frames 60 r4.bin r5.bin > $O/ab_c3.log 2>&1; tail -4 $O/ab_c3.log
AB_BAND=59,76 timeout 300 python tools/ab_fast.py --rounds 3 r4.bin r5.bin > $O/ab_c3_band.log 2>&1; tail -4 $O/ab_c3_band.log
