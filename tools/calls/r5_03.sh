#!/bin/bash
# call 3: occlusion culling in the painters — parity (culling on, strips forced too), A/B per configuration
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r5_03; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
FORMA_HIP_DEBUG=strip_tiles=100000000 timeout 900 python -m pytest tests -m gpu -x -q -k "not multi and not exchange" > $O/pytest_strips.log 2>&1; echo "pytest(strips forced) rc $?"; tail -3 $O/pytest_strips.log
export AB_KERNELS=1
C=cull.bin@FORMA_HIP_DEBUG
timeout 300 python tools/ab_fast.py --workload cubics-1080p --rounds 2 $C=no_cull,strip_tiles=0 $C=strip_tiles=0 $C=strip_tiles=100000 > $O/ab_c2.log 2>&1; cat $O/ab_c2.log
timeout 300 python tools/ab_fast.py --rounds 2 $C=no_cull $C=strip_tiles=0 > $O/ab_c3.log 2>&1; cat $O/ab_c3.log
AB_BAND=59,76 timeout 300 python tools/ab_fast.py --rounds 2 $C=no_cull,strip_tiles=0 $C=strip_tiles=0 $C=strip_tiles=100000 > $O/ab_c3_band.log 2>&1; cat $O/ab_c3_band.log
timeout 300 python tools/ab_fast.py --workload triangles-10m-8k --rounds 1 $C=no_cull $C=strip_tiles=0 > $O/ab_c4.log 2>&1; cat $O/ab_c4.log
timeout 300 python tools/ab_fast.py --workload circles-20k --rounds 1 $C=no_cull,strip_tiles=0 $C=strip_tiles=0 $C=strip_tiles=100000 > $O/ab_circ.log 2>&1; cat $O/ab_circ.log
