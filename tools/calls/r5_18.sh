#!/bin/bash
# call 18: culling only once deep tiles were seen; the digit passes on fewer CUs with frame slots
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r5_18; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
FORMA_HIP_DEBUG=force_cull timeout 600 python -m pytest tests -m gpu -x -q -k "not full_size and not three_frames and not svg_loader and not 10m" > $O/pytest_cull.log 2>&1; echo "pytest(force_cull) rc $?"; tail -2 $O/pytest_cull.log
N=new.bin@FORMA_HIP_DEBUG
timeout 500 python tools/ab_fast.py --rounds 3 --frames 60 r4.bin $N=sort_cus=0 $N=sort_cus=192 new.bin $N=sort_cus=128 > $O/ab_c3.log 2>&1; tail -8 $O/ab_c3.log
timeout 300 python tools/ab_fast.py --workload cubics-1080p --rounds 2 r4.bin $N=sort_cus=0 new.bin > $O/ab_c2.log 2>&1; tail -5 $O/ab_c2.log
