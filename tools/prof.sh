#!/bin/bash
# tools/prof.sh TAG [extra bench args]: rocprofv3 kernel trace of a short bench run -> gpurun_out/TAG/, prints per-kernel averages
TAG=${1:-prof}; shift
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
(cd $OLDPWD && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pmc --no-animated "$@" > $OUT/bench.log 2>&1)
cd $OLDPWD
f=$(ls $OUT/*/*kernel_stats.csv 2>/dev/null | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"].split("(")[0].replace("void ", "")[:44]
    print("%-46s calls %5s  avg %9.1f us  %5.1f%%" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
tail -1 $OUT/bench.log | cut -c1-400
