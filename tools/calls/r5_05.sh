#!/bin/bash
# call 5: the 512-lane carry kernel (three workgroups per CU) with 3..6 slices per row; culling with the ballot pick
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r5_05; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
export AB_KERNELS=1
H=half6.bin@FORMA_HIP_DEBUG
timeout 400 python tools/ab_fast.py --rounds 2 $H=carry_half=0 half6.bin $H=carry_half=3 $H=carry_half=4 $H=carry_half=6 half5.bin half5.bin@FORMA_HIP_DEBUG=carry_half=3 > $O/ab_c3.log 2>&1; cat $O/ab_c3.log
timeout 300 python tools/ab_fast.py --workload triangles-10m-8k --rounds 1 $H=carry_half=0 half6.bin half5.bin > $O/ab_c4.log 2>&1; cat $O/ab_c4.log
timeout 300 python tools/ab_fast.py --workload cubics-1080p --rounds 1 $H=carry_half=0 half6.bin half5.bin > $O/ab_c2.log 2>&1; cat $O/ab_c2.log
AB_BAND=59,76 timeout 300 python tools/ab_fast.py --rounds 1 $H=carry_half=0 half6.bin half5.bin > $O/ab_c3_band.log 2>&1; cat $O/ab_c3_band.log
