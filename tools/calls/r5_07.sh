#!/bin/bash
# call 7: heavy-tiles-first order (heavy list + flags), early staging requests of shallow lists
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r5_07; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
export AB_KERNELS=1
L=ord2.bin@FORMA_HIP_DEBUG
timeout 400 python tools/ab_fast.py --rounds 3 --frames 60 $L=no_order ord2.bin nopre.bin > $O/ab_c3.log 2>&1; cat $O/ab_c3.log
timeout 300 python tools/ab_fast.py --workload triangles-10m-8k --rounds 2 $L=no_order ord2.bin nopre.bin > $O/ab_c4.log 2>&1; cat $O/ab_c4.log
timeout 300 python tools/ab_fast.py --workload cubics-1080p --rounds 2 $L=no_order ord2.bin nopre.bin > $O/ab_c2.log 2>&1; cat $O/ab_c2.log
AB_BAND=59,76 timeout 300 python tools/ab_fast.py --rounds 1 ord2.bin nopre.bin > $O/ab_c3_band.log 2>&1; cat $O/ab_c3_band.log
