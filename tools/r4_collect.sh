#!/bin/bash
# copies what tools/r4_profiles.sh left in gpurun_out/r04 into profiles/r04_* (run here, after the gpurun call)
S=gpurun_out/r04; P=profiles
for w in default circles-20k cubics-1080p triangles-10m-8k exchange_world1 multi_4x_one_gpu multi_rccl_world1 driver_form; do cp $S/bench_$w.json $P/r04_bench_$w.json; done
cp $S/prof_default_kernel_stats.csv $P/r04_kernel_stats_default.csv
cp $S/prof_inflight1_kernel_stats.csv $P/r04_kernel_stats_inflight1.csv
cp $S/prof_triangles_kernel_stats.csv $P/r04_kernel_stats_triangles_inflight1.csv
cp $S/pmc_summary.json $P/r04_pmc_summary.json
cp $S/band_proxy_c3.json $P/r04_band_proxy.json; cp $S/band_proxy_c4.json $P/r04_band_proxy_c4.json
cp $S/d2h_bench.log $P/r04_d2h_bench.txt
