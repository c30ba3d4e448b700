#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench_driver_form.err
for i in 1 2; do timeout 200 python bench.py --no-cpu-baseline --no-animated --no-d2h > $O/bench_repeat_$i.json 2>/dev/null; done
python - <<'PY'
import json
for f in ["bench_default","bench_driver_form","bench_repeat_1","bench_repeat_2"]:
    d=json.loads(open("gpurun_out/r04/%s.json"%f).read().strip().splitlines()[-1])
    print(f, d["value"], d["fps_blocks"], d["fps_render_call"]["median"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"])
PY
