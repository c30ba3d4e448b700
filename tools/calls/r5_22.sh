#!/bin/bash
# call 22: the suite under the hostile switches — poisoned allocations and per-frame buffers, every tile filed as heavy, read-backs on every frame
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r5_22; mkdir -p $O
K="not full_size and not three_frames and not svg_loader and not 10m"
FORMA_HIP_DEBUG=poison=0xFF,poison_frame=0xFF,order_thr=1 timeout 900 python -m pytest tests -m gpu -x -q -k "$K" > $O/pytest_poison_ff.log 2>&1; echo "poison 0xFF + order_thr=1 rc $?"; tail -2 $O/pytest_poison_ff.log
FORMA_HIP_DEBUG=poison=0x00,poison_frame=0x00,force_cull timeout 900 python -m pytest tests -m gpu -x -q -k "$K" > $O/pytest_poison_00.log 2>&1; echo "poison 0x00 + force_cull rc $?"; tail -2 $O/pytest_poison_00.log
FORMA_HIP_DEBUG=sync timeout 900 python -m pytest tests -m gpu -x -q -k "$K" > $O/pytest_sync.log 2>&1; echo "sync rc $?"; tail -2 $O/pytest_sync.log
