#!/bin/bash
# call 10: order with 64 lists per band, limited to launches of a few rounds
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r5_10; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
export AB_KERNELS=1
L=ord5.bin@FORMA_HIP_DEBUG
timeout 300 python tools/ab_fast.py --rounds 2 --frames 60 $L=no_order ord5.bin > $O/ab_c3.log 2>&1; tail -4 $O/ab_c3.log
timeout 300 python tools/ab_fast.py --workload triangles-10m-8k --rounds 1 $L=no_order ord5.bin > $O/ab_c4.log 2>&1; tail -4 $O/ab_c4.log
AB_BAND=224,288 timeout 300 python tools/ab_fast.py --workload triangles-10m-8k --rounds 2 $L=no_order ord5.bin > $O/ab_c4_band.log 2>&1; tail -4 $O/ab_c4_band.log
timeout 300 python tools/ab_fast.py --workload cubics-1080p --rounds 2 $L=no_order ord5.bin > $O/ab_c2.log 2>&1; tail -4 $O/ab_c2.log
timeout 300 python tools/ab_fast.py --workload circles-20k --rounds 1 $L=no_order ord5.bin > $O/ab_circ.log 2>&1; tail -4 $O/ab_circ.log
