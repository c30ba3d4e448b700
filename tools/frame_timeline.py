"""Print the kernel timeline of the last full frame in a rocprofv3 --kernel-trace CSV (start offset, kernel, duration in us)."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [(r["Kernel_Name"].split("(")[0][:40], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000) for r in rows]
idx = [i for i, (n, _) in enumerate(names) if n.startswith("k_line_len")]
s, e = idx[-2], idx[-1]
t0 = int(rows[s]["Start_Timestamp"])
for i in range(s, e):
    print("%8.1f %-42s %7.1f" % ((int(rows[i]["Start_Timestamp"]) - t0) / 1000, names[i][0], names[i][1]))
