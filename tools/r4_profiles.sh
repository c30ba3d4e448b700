#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
bash tools/round_profiles.sh r04 > gpurun_out/r04_round.log 2>&1
O=gpurun_out/r04
python tools/ab_fast.py --rounds 0 > /dev/null 2>&1
python tools/ab_fast.py --workload triangles-10m-8k --rounds 0 > /dev/null 2>&1
timeout 600 python tools/band_proxy.py --slots 1,3,4 --frames 300 --out $O/band_proxy_c3.json > $O/band_proxy_c3.log 2>&1
timeout 600 python tools/band_proxy.py --workload triangles-10m-8k --slots 1,3,4 --frames 300 --out $O/band_proxy_c4.json > $O/band_proxy_c4.log 2>&1
timeout 600 python tools/d2h_bench.py > $O/d2h_bench.log 2>&1
# the bench line exactly as the driver runs it
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench_driver_form.err
tail -c 1500 $O/bench_default.json; echo; tail -2 $O/band_proxy_c3.log; tail -2 $O/band_proxy_c4.log; cat $O/d2h_bench.log | tail -2; ls $O
