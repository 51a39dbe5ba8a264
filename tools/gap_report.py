"""Idle gaps inside each hardware queue's chain of launches, from tools/prof_timeline.py's output: where does a chain wait, and
for how long?  usage: python tools/gap_report.py <timeline.txt> [min gap in us, default 10]"""
import sys
lines = open(sys.argv[1]).read().splitlines()
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
print(lines[0])
rows = []
for l in lines[1:]:
    p = l.split()
    rows.append((float(p[0]), float(p[3]), p[4], p[6] if len(p) > 6 else ""))
for q in sorted({r[2] for r in rows}):
    r = [x for x in rows if x[2] == q]
    busy = sum(x[1] for x in r)
    print(f"{q}: {len(r)} launches, busy {busy:.1f} us, first at {r[0][0]:.1f} ({r[0][3][:36]}), last ends {r[-1][0] + r[-1][1]:.1f}")
    tot = 0.0
    for a, b in zip(r, r[1:]):
        gap = b[0] - (a[0] + a[1])
        if gap > thr:
            tot += gap
            print(f"   gap {gap:7.1f} us at {a[0] + a[1]:8.1f}: {a[3][:36]:36s} -> {b[3][:36]}")
    print(f"   gaps > {thr:g} us: {tot:.1f} us")
