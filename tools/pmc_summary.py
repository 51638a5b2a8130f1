"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel and counter, the per-dispatch value (summed over
the rows rocprofv3 emits per dispatch: XCDs / instances), averaged and max over dispatches.
usage: python tools/pmc_summary.py <dir with *counter_collection.csv> [kernel substring]"""
import csv, glob, os, sys
from collections import defaultdict

root = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else ""
per = defaultdict(float)
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "")
            if filt and filt not in k:
                continue
            per[(k, row["Counter_Name"], f, row["Dispatch_Id"])] += float(row["Counter_Value"])
agg = defaultdict(list)
for (k, c, _, _), v in per.items():
    agg[(k, c)].append(v)
for (k, c), vs in sorted(agg.items()):
    print(f"{c:12s} n={len(vs):4d} avg={sum(vs) / len(vs):16.1f} max={max(vs):16.1f}  {k[:110]}")
