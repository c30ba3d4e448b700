#!/bin/bash
# call 6: painters take the heaviest tiles of the previous frame first
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r5_06; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
export AB_KERNELS=1
L=lpt.bin@FORMA_HIP_DEBUG
timeout 400 python tools/ab_fast.py --rounds 3 $L=no_order lpt.bin > $O/ab_c3.log 2>&1; cat $O/ab_c3.log
timeout 300 python tools/ab_fast.py --workload triangles-10m-8k --rounds 2 $L=no_order lpt.bin > $O/ab_c4.log 2>&1; cat $O/ab_c4.log
timeout 300 python tools/ab_fast.py --workload cubics-1080p --rounds 2 $L=no_order lpt.bin > $O/ab_c2.log 2>&1; cat $O/ab_c2.log
timeout 300 python tools/ab_fast.py --workload circles-20k --rounds 1 $L=no_order lpt.bin > $O/ab_circ.log 2>&1; cat $O/ab_circ.log
