#!/bin/bash
# A/B builds of libforma_hip.so on the SAME box (boxes of the pool differ by 10-25 %):
#   tools/ab_bench.sh ROUNDS a.bin b.bin ...      (files under forma_amd/csrc/variants/)
R=$1; shift
cd forma_amd/csrc; cp libforma_hip.so /tmp/lib_keep.so
for r in $(seq 1 $R); do
  for v in "$@"; do
    cp variants/$v libforma_hip.so
    (cd ../..; timeout 100 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-animated 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); s=d['stages_us']; print('%-14s' % '$v', d['value'], 'pass_us', d['roofline']['avg_launch_us'], {k:round(v) for k,v in s.items()})")
  done
done
cp /tmp/lib_keep.so libforma_hip.so
