#!/bin/bash
# tools/round_profiles.sh TAG: everything profiles/ holds for a round, into gpurun_out/TAG/ (copy what is to be judged into profiles/):
# bench lines of every configuration (+ the line exactly as the driver runs it), the multi-GPU rehearsals on one GPU, rocprofv3
# kernel stats (default, one frame in flight, the 8K scene), the PMC summary, the band proxy with per-kernel times, rocprofv3 traces
# of a one-row and a 1/8 band, the D2H rates.
TAG=${1:-r06}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
Q="--no-cpu-baseline --no-pmc --no-animated"
timeout 400 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_form.json 2> $OUT/bench_driver_form.err
for w in cubics-1080p triangles-10m-8k circles-20k; do
  timeout 200 python bench.py --workload $w $Q > $OUT/bench_$w.json 2>> $OUT/bench_default.err
done
# the multi-GPU paths rehearsed on the one GPU of the box: ONE context over four "devices" (forma_hip_create_multi, device copies
# instead of RCCL), the same with a world of one through RCCL, and the process-per-GPU exchange layout with one rank
# (round 6: `--gpus 4` WITHOUT a launcher = one process over four device contexts; the line lists both layouts, bands and exchange)
FORMA_BENCH_DEVICES=0,0,0,0 timeout 400 python bench.py --gpus 4 $Q > $OUT/bench_gpus4_in_process_one_gpu.json 2>> $OUT/bench_default.err
FORMA_BENCH_MODE_AT_1=1 FORMA_HIP_DEBUG=force_exchange timeout 300 python bench.py $Q > $OUT/bench_multi_rccl_world1.json 2>> $OUT/bench_default.err
FORMA_BENCH_MODE_AT_1=1 timeout 300 python bench.py $Q --mode exchange > $OUT/bench_exchange_world1.json 2>> $OUT/bench_default.err
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_default -- python $OLDPWD/bench.py $Q --no-d2h > $OUT/prof_default.log 2>&1)
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_inflight1 -- python $OLDPWD/bench.py $Q --no-d2h --in-flight 1 > $OUT/prof_inflight1.log 2>&1)
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_triangles -- python $OLDPWD/bench.py $Q --no-d2h --in-flight 1 --workload triangles-10m-8k > $OUT/prof_triangles.log 2>&1)
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_cubics -- python $OLDPWD/bench.py $Q --no-d2h --in-flight 1 --workload cubics-1080p > $OUT/prof_cubics.log 2>&1)
for d in prof_default prof_inflight1 prof_triangles prof_cubics; do cp $OUT/$d/*/*kernel_stats.csv $OUT/${d}_kernel_stats.csv 2>/dev/null; rm -rf $OUT/$d; done
timeout 400 python tools/pmc_round.py $OUT/pmc_summary.json > $OUT/pmc.log 2>&1
# bands: what one device of an 8-GPU context runs per frame (scene tables parked by ab_fast.py)
python tools/ab_fast.py --rounds 0 > /dev/null 2>&1
python tools/ab_fast.py --workload triangles-10m-8k --rounds 0 > /dev/null 2>&1
timeout 600 python tools/band_proxy.py --slots 1,3 --frames 300 --out $OUT/band_proxy_c3.json > $OUT/band_proxy_c3.log 2>&1
timeout 600 python tools/rank_proxy.py --out $OUT/rank_proxy_c3.json > $OUT/rank_proxy_c3.log 2>&1
timeout 600 python tools/rank_proxy.py --workload triangles-10m-8k --out $OUT/rank_proxy_c4.json > $OUT/rank_proxy_c4.log 2>&1
timeout 600 python tools/band_proxy.py --workload triangles-10m-8k --slots 1,3 --frames 300 --out $OUT/band_proxy_c4.json > $OUT/band_proxy_c4.log 2>&1
for b in 67,68 59,76; do
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_band_$b -- python $OLDPWD/tools/band_trace.py --band $b > $OUT/prof_band_$b.log 2>&1)
  cp $OUT/prof_band_$b/*/*kernel_stats.csv $OUT/band_kernels_rows_${b/,/-}.csv 2>/dev/null; rm -rf $OUT/prof_band_$b
done
timeout 600 python tools/d2h_bench.py > $OUT/d2h_bench.log 2>&1
tail -c 600 $OUT/bench_default.json; echo; tail -3 $OUT/bench_default.err; tail -1 $OUT/band_proxy_c3.log; tail -1 $OUT/band_proxy_c4.log; tail -c 700 $OUT/rank_proxy_c3.log; tail -2 $OUT/d2h_bench.log; ls $OUT
