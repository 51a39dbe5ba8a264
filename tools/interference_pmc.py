#!/usr/bin/env python3
"""Totals of the TCC counters of tools/interference.sh's PMC passes: the two chains alone vs replayed together (under --pmc the
dispatches are serialised, so "together" here means INTERLEAVED in time, not co-resident: it isolates what one chain's kernels
do to the other's cache contents from what sharing the CUs does).  usage: python tools/interference_pmc.py <outdir>"""
import csv
import glob
import sys
from collections import defaultdict

out = sys.argv[1]
res = {}
for d in sorted(glob.glob(out + "/p?_*")):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not f:
        continue
    mode = d.split("_", 1)[1] if "_" in d.rsplit("/", 1)[1] else d
    mode = d.rsplit("/", 1)[1].split("_", 1)[1]
    tot = defaultdict(float)
    n_adam = 0
    seen = set()
    for r in csv.DictReader(open(f[0])):
        tot[r["Counter_Name"]] += float(r["Counter_Value"])
        if "adam_kernel" in r["Kernel_Name"]:
            key = (r.get("Dispatch_Id"), r["Kernel_Name"])
            if key not in seen:
                seen.add(key); n_adam += 1
    per = {"critic_only": 2, "gen_only": 1, "full": 3}[mode]
    it = max(1.0, n_adam / per)
    for k, v in tot.items():
        res[(mode, k)] = v / it
names = sorted({k for _, k in res})
print(f"{'counter (per iteration)':40s} {'critic alone':>14s} {'gen alone':>14s} {'sum':>14s} {'interleaved':>14s} {'ratio':>7s}")
for k in names:
    a, b, c = res.get(("critic_only", k), 0.0), res.get(("gen_only", k), 0.0), res.get(("full", k), 0.0)
    print(f"{k:40s} {a:14.0f} {b:14.0f} {a + b:14.0f} {c:14.0f} {c / (a + b) if a + b else float('nan'):7.2f}")
