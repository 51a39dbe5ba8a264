#!/usr/bin/env python3
"""What happens to the kernels of the two chains of an iteration when they run side by side.

    python tools/interference.py <critic_only.db> <gen_only.db> <full.db> [--pmc alone_c.csv alone_g.csv both.csv]

The three rocpd databases are rocprofv3 --kernel-trace runs of tools/graph_gap_probe.py {critic_only, gen_only, full}: the same
captured graphs replayed alone and together.  Per kernel name: launches per iteration, average duration alone and together, and
the fraction of its together-time during which another kernel was resident.  If side-by-side execution were free, durations
would not change; if two co-resident kernels simply SHARE the chip (each wants every CU), a kernel's duration grows by about
its co-resident time times the share it loses.  The summary fits that model:  t_together ~ t_alone + beta * t_coresident  and
reports beta (0 = free overlap, 1 = co-resident time is fully serialised: overlap buys nothing)."""
import re
import sqlite3
import sys
from collections import defaultdict


def short(n):
    m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", n)
    if m:                                                       # mangled name of an anonymous-namespace kernel: base + integer template arguments
        ln = int(m.group(1))
        base, rest = n[m.end():m.end() + ln], n[m.end() + ln:]
        ints = re.findall(r"L[ib]n?(\d+)E", rest.split("EEv")[0]) if rest.startswith("I") else []
        ty = "bf16" if "DF16b" in rest[:8] else "f16" if "DF16_" in rest[:8] else "f32" if rest.startswith("If") else ""
        return f"{base}<{ty}{',' if ty and ints else ''}{','.join(ints)}>"[:72]
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    n = re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", n)
    return n[:72]


def load(db):
    """launches of the probe's measured loop only: behind the LAST long pause of the run (graph capture / the synchronize in front
    of the loop) and without the first fifth of what follows (tools/graph_gap_probe.py replays 20 whole iterations first)"""
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end, grid_x*grid_y*grid_z/(workgroup_x*workgroup_y*workgroup_z) from kernels order by start").fetchall()
    cut = 0
    for i in range(1, len(rows)):
        if rows[i][1] - rows[i - 1][2] > 20e6:                   # > 20 ms without a kernel
            cut = i
    rows = rows[cut:]
    t0, t1 = rows[0][1], rows[-1][2]
    return [r for r in rows if r[1] > t0 + 0.2 * (t1 - t0)]


def per_kernel(rows, with_overlap=False):
    acc = defaultdict(lambda: [0, 0.0, 0.0])
    if with_overlap:
        # co-resident time of every launch: sweep over starts / ends
        ev = sorted([(s, 1, i) for i, (_, s, e, _) in enumerate(rows)] + [(e, -1, i) for i, (_, s, e, _) in enumerate(rows)])
        co = [0.0] * len(rows)
        live, last = set(), ev[0][0]
        for t, d, i in ev:
            if len(live) >= 2:
                for j in live:
                    co[j] += t - last
            last = t
            if d == 1: live.add(i)
            else: live.discard(i)
    for i, (name, s, e, wg) in enumerate(rows):
        k = (short(name), int(wg))
        a = acc[k]
        a[0] += 1; a[1] += (e - s) / 1e3
        if with_overlap:
            a[2] += co[i] / 1e3
    return acc


def main():
    dbs = [a for a in sys.argv[1:4]]
    alone = {}
    iters = {}
    for tag, db in zip(("critic", "gen"), dbs[:2]):
        rows = load(db)
        pk = per_kernel(rows)
        n_it = max(1, sum(v[0] for k, v in pk.items() if "adam_kernel" in k[0]) / (2 if tag == "critic" else 1))
        iters[tag] = n_it
        for k, v in pk.items():
            if v[0] / n_it < 0.5:                                # (a straggler of the warm-up, not a launch of this chain)
                continue
            if k in alone:                                       # a launch shape both chains have: launch-weighted mean
                t0, p0, a0 = alone[k]
                alone[k] = ("both", p0 + v[0] / n_it, (a0 * p0 + v[1] / n_it) / (p0 + v[0] / n_it))
            else:
                alone[k] = (tag, v[0] / n_it, v[1] / v[0])
    rows = load(dbs[2])
    pk = per_kernel(rows, with_overlap=True)
    n_it = max(1, sum(v[0] for k, v in pk.items() if "adam_kernel" in k[0]) / 3)
    print(f"iterations: critic-only {iters['critic']:.0f}, generator-only {iters['gen']:.0f}, together {n_it:.0f}")
    tot = dict(alone=0.0, together=0.0, co=0.0)
    table = []
    for k, v in pk.items():
        if k not in alone:
            continue
        tag, per_it, t_alone = alone[k]
        t_tog, co = v[1] / v[0], v[2] / v[0]
        # a kernel name + grid that BOTH chains launch (clip/adam, prep, ...) is attributed to the last one loaded: fine for totals
        table.append((v[1] / n_it, k, tag, v[0] / n_it, t_alone, t_tog, co))
        tot["alone"] += t_alone * v[0] / n_it; tot["together"] += v[1] / n_it; tot["co"] += v[2] / n_it
    table.sort(reverse=True)
    print(f"{'kernel (workgroups)':64s} chain  /iter   alone us  together us  co-resident us  inflation")
    for _, k, tag, per_it, ta, tt, co in table[:28]:
        print(f"{k[0][:56]:56s} {k[1]:6d} {tag:6s} {per_it:5.1f}  {ta:8.1f}  {tt:10.1f}  {co:12.1f}  {tt / ta:8.2f}")
    # least squares through the origin of (t_together - t_alone) on t_coresident, weighted by launches
    num = sum(w * (tt - ta) * co for w, k, tag, per_it, ta, tt, co in table)
    den = sum(w * co * co for w, k, tag, per_it, ta, tt, co in table)
    beta = num / den if den else float("nan")
    print(f"\nkernel-busy per iteration: alone {tot['alone'] / 1e3:.3f} ms, together {tot['together'] / 1e3:.3f} ms "
          f"(+{(tot['together'] / tot['alone'] - 1) * 100:.1f} %), of which co-resident {tot['co'] / 1e3:.3f} ms")
    print(f"fit  t_together = t_alone + beta * t_coresident:  beta = {beta:.2f}   (0: overlap is free; 0.5: two co-resident kernels "
          f"each run at half speed, i.e. the chip's throughput is merely shared; 1: co-resident time is lost)")


if __name__ == "__main__":
    main()
