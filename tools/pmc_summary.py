#!/usr/bin/env python3
"""Per-kernel averages of the counters of one rocprofv3 --pmc pass (csv output): python tools/pmc_summary.py DIR [filter]"""
import csv
import glob
import re
import sys
from collections import defaultdict

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
if not f:
    sys.exit("no counter_collection.csv under " + d)
acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"]
    if flt and flt not in k:
        continue
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = defaultdict(list)
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
if kt:
    for r in csv.DictReader(open(kt[0])):
        dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, cs in sorted(acc.items(), key=lambda kv: -sum(dur.get(kv[0], [0]))):
    name = re.sub(r"\(anonymous namespace\)::|void ", "", k)[:90]
    ds = dur.get(k, [])
    ds2 = ds[len(ds) // 2:] if ds else []
    print(f"{name}  launches {len(ds)}  avg_us(last half) {sum(ds2) / max(len(ds2), 1):.1f}")
    for c, v in cs.items():
        v2 = v[len(v) // 2:]
        print(f"    {c:32s} {sum(v2) / len(v2):16.1f}")
