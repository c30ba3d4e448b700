#!/bin/bash
# tools/ab_prof.sh KERNEL_SUBSTR a.bin b.bin ...: per-variant average duration of one kernel (rocprofv3 kernel trace of a short
# single-frame-in-flight bench run), same box
K=$1; shift
cp forma_amd/csrc/libforma_hip.so /tmp/lib_keep.so
export TMPDIR=/tmp
for v in "$@"; do
  cp forma_amd/csrc/variants/$v forma_amd/csrc/libforma_hip.so
  rm -rf /tmp/abp_$v
  (cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abp_$v -- python $OLDPWD/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pmc --no-animated --in-flight 1 > /tmp/abp_$v.log 2>&1)
  f=$(ls /tmp/abp_$v/*/*kernel_stats.csv | head -1)
  python - "$f" "$K" "$v" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Name"]:
        print("%-14s %-30s avg %8.1f us" % (sys.argv[3], r["Name"].split("(")[0].replace("void ", "")[:30], float(r["AverageNs"]) / 1e3))
PY
  tail -1 /tmp/abp_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   fps', d['value'])"
done
cp /tmp/lib_keep.so forma_amd/csrc/libforma_hip.so
