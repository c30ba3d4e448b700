#!/bin/bash
# call 12: segment window (one load of a tile's painted segments), new tests
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r5_12; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest.log
FORMA_HIP_DEBUG=strip_tiles=100000000 timeout 900 python -m pytest tests -m gpu -x -q -k "not multi and not exchange" > $O/pytest_strips.log 2>&1; echo "pytest(strips forced) rc $?"; tail -3 $O/pytest_strips.log
export AB_KERNELS=1
timeout 300 python tools/ab_fast.py --rounds 3 --frames 60 nowin.bin win.bin > $O/ab_c3.log 2>&1; tail -5 $O/ab_c3.log; grep kernels $O/ab_c3.log | tail -2
timeout 300 python tools/ab_fast.py --workload triangles-10m-8k --rounds 2 nowin.bin win.bin > $O/ab_c4.log 2>&1; tail -5 $O/ab_c4.log; grep kernels $O/ab_c4.log | tail -2
AB_BAND=59,76 timeout 300 python tools/ab_fast.py --rounds 2 nowin.bin win.bin > $O/ab_c3_band.log 2>&1; tail -5 $O/ab_c3_band.log; grep kernels $O/ab_c3_band.log | tail -2
timeout 300 python tools/ab_fast.py --workload circles-20k --rounds 1 nowin.bin win.bin > $O/ab_circ.log 2>&1; tail -4 $O/ab_circ.log
timeout 300 python tools/ab_fast.py --workload cubics-1080p --rounds 1 nowin.bin win.bin > $O/ab_c2.log 2>&1; tail -4 $O/ab_c2.log
