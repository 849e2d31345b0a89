"""Reduce a rocprofv3 --pmc output directory to a small per-kernel summary (mean counter value per dispatch) for the
kvc:: kernels, so it fits in profiles/.   python tools/pmc_summary.py <dir> <out.csv>"""
import csv, glob, os, sys
from collections import defaultdict
d, out = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "")
            if "kvc::" not in name:
                continue
            short = name.split("(")[0].replace("void ", "")
            key = (short, row["Counter_Name"])
            acc[key][0] += float(row["Counter_Value"])
            acc[key][1] += 1
with open(out, "w", newline="") as fh:
    w = csv.writer(fh, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["kernel", "counter", "mean_per_dispatch", "dispatches"])
    for (k, c), (s, n) in sorted(acc.items()):
        w.writerow([k, c, round(s / n, 1), n])
print(open(out).read())
