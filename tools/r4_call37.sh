#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4final; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke2.log 2>&1; tail -1 $O/smoke2.log
timeout 900 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/pytest_default2.log 2>&1; tail -1 $O/pytest_default2.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-animated 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['fps_blocks'], d['fps_render_call']['median'], d['roofline']['avg_launch_us'], d['roofline']['frac'])"
