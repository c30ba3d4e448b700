#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4o; mkdir -p $O
export PYTHONUNBUFFERED=1
python tools/ab_fast.py --rounds 0 > /dev/null 2>&1
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace -- python tools/d2h_trace.py > $O/trace.log 2>&1
ls -R $O/trace | head -20
python - <<'PY'
import csv, glob
ks = glob.glob("gpurun_out/r4o/trace/**/*kernel_trace.csv", recursive=True)
ms = glob.glob("gpurun_out/r4o/trace/**/*memory_copy_trace.csv", recursive=True)
ev = []
for f in ks:
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:28]))
for f in ms:
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") ))
ev.sort()
# the last frame: from the last k_line_len on
idx = max(i for i, e in enumerate(ev) if e[2].startswith("k_line_len"))
t0 = ev[idx][0]
for s, e, n in ev[idx:]:
    print("%9.1f %9.1f  %7.1f  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n))
PY
