"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel: mean counter value per launch."""
import csv, sys, collections, glob
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for fn in sys.argv[1:]:
    for f in glob.glob(fn, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:40]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in sorted(acc):
    print(k, {c: round(acc[k][c] / cnt[k][c]) for c in sorted(acc[k])}, "launches", max(cnt[k].values()))
