"""One rocprofv3 --pmc pass of a short single-frame-in-flight bench run, per-kernel means of the frame-sized dispatches:
    python tools/pmc_pass.py COUNTER [COUNTER ...]        (on a GPU box; at most 8 SQ counters per pass)"""
import collections, csv, glob, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
counters = sys.argv[1:]
with tempfile.TemporaryDirectory(dir="/tmp") as d:
    cmd = ["rocprofv3", "--pmc"] + counters + ["--kernel-trace", "--output-format", "csv", "-d", d, "--",
           "python", os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--in-flight", "1", "--no-cpu-baseline", "--no-pmc", "--no-animated"]
    r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[k][row["Counter_Name"]].append((int(row["Grid_Size"]), float(row["Counter_Value"])))
    if not acc:
        print(r.stdout[-2000:], r.stderr[-2000:])
    for k, cs in acc.items():
        if k.startswith("__amd"):
            continue
        out = []
        for c in counters:
            v = cs.get(c, [])
            if v:
                big = max(g for g, _ in v)
                vals = [x for g, x in v if g == big]
                out.append("%s %.4g" % (c, sum(vals) / len(vals)))
        print("%-22s" % k[:22], "  ".join(out))
