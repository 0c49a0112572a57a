#!/usr/bin/env python3
"""Per-kernel means of every counter found in the given rocprofv3 counter_collection.csv files."""
import collections, csv, re, sys
d = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for path in sys.argv[1:]:
    try:
        rows = csv.DictReader(open(path))
    except OSError:
        continue
    for r in rows:
        name = re.sub(r"<.*", "", r["Kernel_Name"].replace("void ", "")).replace("d3m::", "").split("(")[0]
        if not name.startswith("k_"):
            continue
        c = d[name][r["Counter_Name"]]
        c[0] += 1
        c[1] += float(r["Counter_Value"])
names = sorted({c for k in d for c in d[k]})
print("kernel," + ",".join(names))
for k in sorted(d):
    print(k + "," + ",".join(f"{d[k][c][1] / d[k][c][0]:.0f}" if d[k][c][0] else "" for c in names))
