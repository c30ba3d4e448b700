#!/bin/bash
# tools/round_profiles.sh TAG: everything profiles/ holds for a round, into gpurun_out/TAG/ (copy what is to be judged into profiles/)
TAG=${1:-r03}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
Q="--no-cpu-baseline --no-animated"
timeout 400 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
for w in cubics-1080p triangles-10m-8k circles-20k; do
  timeout 200 python bench.py --workload $w $Q > $OUT/bench_$w.json 2>> $OUT/bench_default.err
done
# the multi-GPU paths rehearsed on the one GPU of the box: ONE context over four "devices" (forma_hip_create_multi, device copies
# instead of RCCL), the same with a world of one through RCCL, and the process-per-GPU exchange layout with one rank
FORMA_BENCH_MODE_AT_1=1 FORMA_BENCH_DEVICES=0,0,0,0 timeout 300 python bench.py $Q > $OUT/bench_multi_4x_one_gpu.json 2>> $OUT/bench_default.err
FORMA_BENCH_MODE_AT_1=1 FORMA_HIP_DEBUG=force_exchange timeout 300 python bench.py $Q > $OUT/bench_multi_rccl_world1.json 2>> $OUT/bench_default.err
FORMA_BENCH_MODE_AT_1=1 timeout 300 python bench.py $Q --mode exchange > $OUT/bench_exchange_world1.json 2>> $OUT/bench_default.err
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_default -- python $OLDPWD/bench.py $Q --no-d2h > $OUT/prof_default.log 2>&1)
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_inflight1 -- python $OLDPWD/bench.py $Q --no-d2h --in-flight 1 > $OUT/prof_inflight1.log 2>&1)
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_triangles -- python $OLDPWD/bench.py $Q --no-d2h --in-flight 1 --workload triangles-10m-8k > $OUT/prof_triangles.log 2>&1)
for d in prof_default prof_inflight1 prof_triangles; do cp $OUT/$d/*/*kernel_stats.csv $OUT/${d}_kernel_stats.csv 2>/dev/null; rm -rf $OUT/$d; done
timeout 400 python tools/pmc_round.py $OUT/pmc_summary.json > $OUT/pmc.log 2>&1
tail -c 600 $OUT/bench_default.json; echo; tail -3 $OUT/bench_default.err
