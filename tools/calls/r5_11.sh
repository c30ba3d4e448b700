#!/bin/bash
# call 11: quad painter (four tiles per wavefront, all-solid scenes)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r5_11; mkdir -p $O
FORMA_HIP_DEBUG=paint_quad=2 timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_quad.log 2>&1; echo "pytest(quad forced) rc $?"; tail -5 $O/pytest_quad.log
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
export AB_KERNELS=1
L=quad.bin@FORMA_HIP_DEBUG
timeout 300 python tools/ab_fast.py --workload triangles-10m-8k --rounds 2 $L=paint_quad=0 quad.bin > $O/ab_c4.log 2>&1; tail -8 $O/ab_c4.log
AB_BAND=224,288 timeout 300 python tools/ab_fast.py --workload triangles-10m-8k --rounds 2 $L=paint_quad=0 quad.bin > $O/ab_c4_band.log 2>&1; tail -8 $O/ab_c4_band.log
timeout 300 python tools/ab_fast.py --workload cubics-1080p --rounds 2 $L=paint_quad=0 quad.bin $L=paint_quad=2 > $O/ab_c2.log 2>&1; tail -10 $O/ab_c2.log
timeout 300 python tools/ab_fast.py --workload circles-20k --rounds 1 $L=paint_quad=0 quad.bin $L=paint_quad=2 > $O/ab_circ.log 2>&1; tail -8 $O/ab_circ.log
