#!/usr/bin/env python3
"""Folds two rocprofv3 --pmc runs (FETCH_SIZE and WRITE_SIZE, collected in SEPARATE passes as
/opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE needs 3 of the 4 TCC slots) into per-kernel
HBM traffic per launch.  Units: the counters are in KiB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B,
so the fetch side is doubled (that guide, section HBM) -- calibrated there for wide coalesced streams, an
upper-bound style estimate for the narrow gathers some of these kernels do.

usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>"""
import collections
import csv
import json
import re
import sys


def fold(path, counter):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"<.*", "", r["Kernel_Name"].replace("void ", "")).replace("d3m::", "").split("(")[0]
        d[name][0] += 1
        d[name][1] += float(r["Counter_Value"])
    return d


def main():
    f, w = fold(sys.argv[1], "FETCH_SIZE"), fold(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in f:
        if not k.startswith("k_"):
            continue
        fetch_kib = f[k][1] / f[k][0]
        write_kib = w[k][1] / w[k][0] if k in w and w[k][0] else 0.0
        out[k] = {"launches": f[k][0], "fetch_size_KiB_raw": round(fetch_kib, 1), "write_size_KiB": round(write_kib, 1),
                  "hbm_bytes_per_launch": int((2 * fetch_kib + write_kib) * 1024)}
    json.dump({"note": "bench.py --steps 3 --warmup 1 --no-graph, 32 views, 100352 tris, 512^2; hbm = 2*FETCH_SIZE + WRITE_SIZE",
               "kernels": out}, open(sys.argv[3], "w"), indent=1)
    for k in sorted(out, key=lambda k: -out[k]["hbm_bytes_per_launch"]):
        print(f"{k:32s} {out[k]['hbm_bytes_per_launch'] / 1e6:10.1f} MB/launch")


if __name__ == "__main__":
    main()
