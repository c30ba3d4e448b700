#!/bin/bash
# tools/ab_prof_all.sh [WORKLOAD] a.bin b.bin b.bin@FORMA_HIP_DEBUG=x ...: rocprofv3 average of EVERY frame kernel per variant
# (short bench run, one frame in flight, no PCIe legs), same box — which kernel moved when a stage time moved
W=paris-like-30k-4k
case "$1" in *.bin*) ;; *) W=$1; shift;; esac
export TMPDIR=/tmp
for spec in "$@"; do
  v=${spec%%@*}; e=""; [ "$spec" != "$v" ] && e=${spec#*@}
  rm -rf /tmp/abpa
  (cd /tmp && env FORMA_HIP_LIB=$OLDPWD/forma_amd/csrc/variants/$v $e timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abpa -- \
     python $OLDPWD/bench.py --workload $W --steps 20 --warmup 3 --no-cpu-baseline --no-pmc --no-animated --no-d2h --in-flight 1 > /tmp/abpa.log 2>&1)
  f=$(ls /tmp/abpa/*/*kernel_stats.csv | head -1)
  python - "$f" "$spec" <<'PY'
import csv, sys
rows = [(r["Name"].split("(")[0].replace("void ", "")[:28], int(r["Calls"]), float(r["AverageNs"]) / 1e3) for r in csv.DictReader(open(sys.argv[1]))]
rows = [r for r in rows if r[1] > 50 and not r[0].startswith("__amd")]
print("%-40s" % sys.argv[2][:40], " ".join("%s %.1f" % (n.replace("k_", ""), a) for n, c, a in sorted(rows)), " sum %.1f" % sum(a * (2 if "onesweep" in n else 1) for n, c, a in rows))
PY
done
