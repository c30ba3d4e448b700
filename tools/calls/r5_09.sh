#!/bin/bash
# call 9: order auto-off on flat scenes; the bench line as the driver runs it; band traces
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r5_09; mkdir -p $O
export AB_KERNELS=1
L=ord4.bin@FORMA_HIP_DEBUG
timeout 300 python tools/ab_fast.py --workload triangles-10m-8k --rounds 2 $L=no_order ord4.bin > $O/ab_c4.log 2>&1; cat $O/ab_c4.log
timeout 300 python tools/ab_fast.py --rounds 1 --frames 60 $L=no_order ord4.bin > $O/ab_c3.log 2>&1; tail -4 $O/ab_c3.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench_driver_form.err; tail -c 2500 $O/bench_driver_form.json; tail -3 $O/bench_driver_form.err
timeout 600 python tools/band_proxy.py --slots 1,3 --frames 300 --out $O/band_proxy_c3.json > $O/band_proxy_c3.log 2>&1; tail -2 $O/band_proxy_c3.log
for b in 67,68 59,76; do
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_band_$b -- python $GRAFT_REPO_ROOT/tools/band_trace.py --band $b > $GRAFT_REPO_ROOT/$O/prof_band_$b.log 2>&1)
  cp $O/prof_band_$b/*/*kernel_stats.csv $O/band_kernels_$b.csv 2>/dev/null; rm -rf $O/prof_band_$b
  head -14 $O/band_kernels_$b.csv | cut -c1-150
done
