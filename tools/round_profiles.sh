#!/bin/bash
# tools/round_profiles.sh TAG: everything profiles/ holds for a round, into gpurun_out/TAG/ (copy what is to be judged into profiles/)
TAG=${1:-r02}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python bench.py --animated > $OUT/bench_default.json 2> $OUT/bench_default.err
for w in cubics-1080p triangles-10m-8k circles-20k; do
  timeout 200 python bench.py --workload $w --no-cpu-baseline > $OUT/bench_$w.json 2>> $OUT/bench_default.err
done
FORMA_BENCH_MODE_AT_1=1 timeout 200 python bench.py --no-cpu-baseline > $OUT/bench_exchange_world1.json 2>> $OUT/bench_default.err
FORMA_BENCH_MODE_AT_1=1 timeout 200 python bench.py --no-cpu-baseline --workload triangles-10m-8k > $OUT/bench_exchange_world1_triangles.json 2>> $OUT/bench_default.err
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_default -- python $OLDPWD/bench.py --no-cpu-baseline > $OUT/prof_default.log 2>&1)
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_inflight1 -- python $OLDPWD/bench.py --no-cpu-baseline --in-flight 1 > $OUT/prof_inflight1.log 2>&1)
for d in prof_default prof_inflight1; do cp $OUT/$d/*/*kernel_stats.csv $OUT/${d}_kernel_stats.csv 2>/dev/null; rm -rf $OUT/$d; done
timeout 300 python tools/pmc_round.py $OUT/pmc_summary.json > $OUT/pmc.log 2>&1
tail -c 600 $OUT/bench_default.json; echo; tail -3 $OUT/bench_default.err
