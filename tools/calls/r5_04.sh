#!/bin/bash
# call 4: culling without the occluder table (painter finds the occluder, group lists culled per group)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r5_04; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
export AB_KERNELS=1
C=cull3.bin@FORMA_HIP_DEBUG
timeout 300 python tools/ab_fast.py --workload cubics-1080p --rounds 2 $C=no_cull cull.bin cull3.bin > $O/ab_c2.log 2>&1; cat $O/ab_c2.log
timeout 300 python tools/ab_fast.py --rounds 2 $C=no_cull cull.bin cull3.bin > $O/ab_c3.log 2>&1; cat $O/ab_c3.log
AB_BAND=59,76 timeout 300 python tools/ab_fast.py --rounds 1 $C=no_cull,strip_tiles=0 cull3.bin > $O/ab_c3_band.log 2>&1; cat $O/ab_c3_band.log
for w in paris-like-30k-4k cubics-1080p; do FORMA_HIP_LIB=$PWD/forma_amd/csrc/variants/prof3.bin timeout 200 python tools/paint_prof.py $w > $O/pprof_$w.log 2>&1; cat $O/pprof_$w.log; done
