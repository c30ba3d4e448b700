#!/bin/bash
# call 1: suite on the timing refactor, per-kernel times of every configuration, tile depth statistics
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r5_01; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
export AB_KERNELS=1
timeout 300 python tools/ab_fast.py --rounds 1 fin.bin base.bin > $O/ab_c3.log 2>&1; cat $O/ab_c3.log
timeout 300 python tools/ab_fast.py --workload cubics-1080p --rounds 1 base.bin > $O/ab_c2.log 2>&1; cat $O/ab_c2.log
timeout 300 python tools/ab_fast.py --workload triangles-10m-8k --rounds 1 base.bin > $O/ab_c4.log 2>&1; cat $O/ab_c4.log
AB_BAND=59,76 timeout 300 python tools/ab_fast.py --rounds 1 base.bin > $O/ab_c3_band.log 2>&1; cat $O/ab_c3_band.log
AB_BAND=67,68 timeout 300 python tools/ab_fast.py --rounds 1 base.bin > $O/ab_c3_band1.log 2>&1; cat $O/ab_c3_band1.log
timeout 120 python tools/tile_stats.py > $O/tiles_c3.log 2>&1; cat $O/tiles_c3.log
timeout 120 python tools/tile_stats.py cubics-1080p > $O/tiles_c2.log 2>&1; cat $O/tiles_c2.log
timeout 120 python tools/tile_stats.py triangles-10m-8k > $O/tiles_c4.log 2>&1; cat $O/tiles_c4.log
for w in paris-like-30k-4k cubics-1080p; do FORMA_HIP_LIB=$PWD/forma_amd/csrc/variants/prof.bin timeout 200 python tools/paint_prof.py $w > $O/pprof_$w.log 2>&1; cat $O/pprof_$w.log; done
timeout 300 python bench.py --no-cpu-baseline --no-animated > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json
