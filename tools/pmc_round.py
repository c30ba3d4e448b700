"""Per-kernel counter summary of one build, for profiles/ and bench.py (roofline.traffic, roofline_painter):

    python tools/pmc_round.py OUT.json          (on a GPU box; runs three rocprofv3 --pmc passes of a short bench.py run)

Passes (SQ has 8 slots per pass, FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950 — MI355X guide, rocprofv3 PMC slots):
  1. SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES SQ_BUSY_CYCLES
  2. FETCH_SIZE        3. WRITE_SIZE
FETCH_SIZE is doubled (gfx950 counts 128-byte requests as 64 bytes: guide, HBM section; checked here on k_runs_count,
whose only traffic is one read of the 8 N byte stream).  Values are means over the frame-sized dispatches of a kernel."""
import collections, csv, glob, json, os, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = ["python", os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--in-flight", "1", "--no-cpu-baseline", "--no-animated", "--no-d2h", "--no-pmc"]
PASSES = [["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_INST_LDS", "SQ_WAVES", "SQ_BUSY_CYCLES"],
          ["FETCH_SIZE"], ["WRITE_SIZE"]]


def run_pass(counters, d, timeout=None):
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--pmc"] + counters + ["--kernel-trace", "--output-format", "csv", "-d", d, "--"] + BENCH
    out = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    return json.loads(line[-1]) if line else None


def per_kernel(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[k][r["Counter_Name"]].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    out = {}
    for k, cs in acc.items():
        out[k] = {}
        for c, v in cs.items():
            big = max(g for g, _ in v)                       # the frame-sized launches only
            vals = [x for g, x in v if g == big]
            out[k][c] = sum(vals) / len(vals)
            out[k]["dispatches"] = max(out[k].get("dispatches", 0), len(vals))   # (how often the kernel ran at frame size: once = the synchronous first frame)
    return out


def collect(passes, timeout=None):
    """one rocprofv3 --pmc run of the short bench command per counter list -> ({kernel: {counter: mean per frame-sized dispatch}}, the run's bench line)"""
    kern, bench_line = collections.defaultdict(dict), None
    for counters in passes:
        with tempfile.TemporaryDirectory(dir="/tmp") as d:
            line = run_pass(counters, d, timeout)
            bench_line = bench_line or line
            for k, v in per_kernel(d).items():
                if not k.startswith("__amd"):
                    kern[k].update(v)
    for k, v in kern.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            v["hbm_bytes_per_launch"] = int(v["FETCH_SIZE"] * 1024 * 2 + v["WRITE_SIZE"] * 1024)
    return kern, bench_line


def main():
    dst = sys.argv[1]
    kern, bench_line = collect(PASSES)
    n = bench_line["config"]["pixel_segments"] if bench_line else 0
    for k, v in kern.items():
        for c in list(v):
            v[c] = round(v[c], 1) if isinstance(v[c], float) else v[c]

    cal = kern.get("k_runs_count", {}).get("FETCH_SIZE", 0) * 1024 * 2 / (8.0 * n) if n else 0
    json.dump({"_note": "rocprofv3 --pmc, three passes of `bench.py --steps 3 --warmup 1 --in-flight 1 --no-cpu-baseline`; mean over the "
                        "frame-sized dispatches; FETCH_SIZE / WRITE_SIZE in KB per dispatch, hbm_bytes_per_launch = 2 x FETCH + WRITE "
                        "(gfx950 FETCH_SIZE correction; calibration on k_runs_count, one read of the 8N-byte stream: measured / expected "
                        "= %.3f); SQ_* are sums over all waves of a dispatch" % cal,
               "n_segments": n, "kernels": kern}, open(dst, "w"), indent=1)
    print("wrote", dst, "calibration", round(cal, 3))


if __name__ == "__main__":
    main()
