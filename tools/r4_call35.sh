#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
Q="--no-cpu-baseline --no-animated"
FORMA_BENCH_MODE_AT_1=1 FORMA_BENCH_DEVICES=0,0,0,0 timeout 300 python bench.py $Q > $O/bench_multi_4x_one_gpu.json 2> $O/multi.err
FORMA_BENCH_MODE_AT_1=1 FORMA_HIP_DEBUG=force_exchange timeout 300 python bench.py $Q > $O/bench_multi_rccl_world1.json 2>> $O/multi.err
FORMA_BENCH_MODE_AT_1=1 timeout 300 python bench.py $Q --mode exchange > $O/bench_exchange_world1.json 2>> $O/multi.err
for w in cubics-1080p triangles-10m-8k circles-20k; do timeout 200 python bench.py --workload $w $Q > $O/bench_$w.json 2>> $O/multi.err; done
python - <<'PY'
import json
for f in ["bench_multi_4x_one_gpu","bench_multi_rccl_world1","bench_exchange_world1","bench_cubics-1080p","bench_triangles-10m-8k","bench_circles-20k"]:
    d=json.loads(open("gpurun_out/r04/%s.json"%f).read().strip().splitlines()[-1])
    print(f, d["value"], d["fps_blocks"]["median"], d["fps_render_call"]["median"], d.get("fps_including_d2h"), d.get("fps_including_d2h_enqueued_three_buffers"), d["roofline"]["avg_launch_us"], d["roofline"]["frac"], d["mpixel_segments_per_s"], d.get("frames_in_flight"))
PY
tail -3 $O/multi.err
