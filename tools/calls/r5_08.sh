#!/bin/bash
# call 8: order with the 2 x mean floor; SVG route at full size; CPU oracle thread sweep after the flat maps
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r5_08; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
export AB_KERNELS=1
L=ord3.bin@FORMA_HIP_DEBUG
timeout 400 python tools/ab_fast.py --rounds 2 --frames 60 $L=no_order ord3.bin > $O/ab_c3.log 2>&1; cat $O/ab_c3.log
timeout 300 python tools/ab_fast.py --workload triangles-10m-8k --rounds 2 $L=no_order ord3.bin > $O/ab_c4.log 2>&1; cat $O/ab_c4.log
timeout 300 python tools/ab_fast.py --workload cubics-1080p --rounds 1 $L=no_order ord3.bin > $O/ab_c2.log 2>&1; cat $O/ab_c2.log
timeout 300 python tools/ab_fast.py --workload circles-20k --rounds 1 $L=no_order ord3.bin > $O/ab_circ.log 2>&1; cat $O/ab_circ.log
timeout 300 python tools/cpu_sweep.py > $O/cpu_sweep.log 2>&1; cat $O/cpu_sweep.log
