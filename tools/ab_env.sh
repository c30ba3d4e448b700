#!/bin/bash
# A/B on the SAME box by environment switch: tools/ab_env.sh "VAR=1" [rounds] [workload]
# prints fps, radix pass us and the per-stage device times with and without the switch
for r in $(seq 1 ${2:-3}); do
  for v in "" "$1"; do
    env $v timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --workload ${3:-paris-like-30k-4k} 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); s=d['stages_us']; print('[%s]' % '$v', d['value'], d['roofline']['avg_launch_us'], {k:round(v) for k,v in s.items()})"
  done
done
