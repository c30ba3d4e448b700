#!/bin/bash
# A/B two builds of libforma_hip.so on the SAME box: tools/ab_bench.sh lib_a.bin lib_b.bin [rounds]
cd forma_amd/csrc
for r in $(seq 1 ${3:-3}); do
  for v in $1 $2; do
    cp $v libforma_hip.so
    (cd ../..; timeout 100 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); s=d['stages_us']; print('$v', d['value'], d['roofline']['avg_launch_us'], {k:round(v) for k,v in s.items()})")
  done
done
