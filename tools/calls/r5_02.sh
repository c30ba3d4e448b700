#!/bin/bash
# call 2: strip painters — parity under forced strips, A/B on C2 / band / C3 / C4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r5_02; mkdir -p $O
FORMA_HIP_DEBUG=strip_tiles=100000000 timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_strips.log 2>&1; echo "pytest(strips forced) rc $?"; tail -3 $O/pytest_strips.log
export AB_KERNELS=1
S=strip.bin@FORMA_HIP_DEBUG=strip_tiles
timeout 300 python tools/ab_fast.py --workload cubics-1080p --rounds 2 $S=0 $S=100000 > $O/ab_c2.log 2>&1; cat $O/ab_c2.log
AB_BAND=59,76 timeout 300 python tools/ab_fast.py --rounds 2 $S=0 $S=100000 > $O/ab_c3_band.log 2>&1; cat $O/ab_c3_band.log
AB_BAND=67,68 timeout 300 python tools/ab_fast.py --rounds 1 $S=0 $S=100000 > $O/ab_c3_band1.log 2>&1; cat $O/ab_c3_band1.log
timeout 300 python tools/ab_fast.py --rounds 1 $S=0 $S=100000 > $O/ab_c3.log 2>&1; cat $O/ab_c3.log
AB_BAND=224,288 timeout 300 python tools/ab_fast.py --workload triangles-10m-8k --rounds 1 $S=0 $S=100000 > $O/ab_c4_band.log 2>&1; cat $O/ab_c4_band.log
