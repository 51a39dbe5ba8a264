#!/usr/bin/env python3
"""Join a rocprofv3 kernel trace of `python bench.py ...` to the conv LABELS of that run (make bench.py's roofline
reproducible from profiles/).

    python tools/prof_labels.py <bench.json> <rocpd results.db> --out profiles/round4_label_durations.json [--csv ...]

bench.json: the JSON line the profiled bench run printed (its `labels` table: per label the kernel template expression the
dispatcher launched -- gcssl_last_kernel -- and the launch's workgroup count -- gcssl_last_grid).  A label's launches in the
trace are the kernel launches whose demangled name starts with that kernel's base name, whose leading integer template
arguments are the ones of the expression, and whose grid has exactly that many workgroups.  Two labels with the same kernel
and grid (D.c2.fwd[n=768] and G.down2.fwd[n=768]) are the same launch shape and share one average.  Launches of the
graph replays dominate the trace (hundreds of iterations against a handful of eager warm-up / probe launches), and the
eager probe launches are filtered out by taking the launches of the REPLAYED region only: those before the last `adam_kernel`
launch that belongs to a run of evenly spaced iterations -- in practice: everything before the probe pass starts, which the
trace shows as the first launch after the longest gap between two launches of the second half of the run."""
import argparse
import json
import re
import sqlite3
from collections import defaultdict
from pathlib import Path


def base_and_ints(expr: str):
    """'(conv_dma_kernel<O, 128, 128, MODE, 4, 2, false, 8>)' -> ('conv_dma_kernel', ['128', '128', None, '4', '2', 'false', '8']):
    the non-type template arguments by POSITION (the first argument of every kernel template here is its element type); an
    argument the launch site spells symbolically (MODE, BM, ...) is None = matches anything."""
    m = re.match(r"\(?\s*([A-Za-z_]\w*)\s*(?:<(.*)>)?\s*\)?$", expr.strip())
    if not m:
        return expr, []
    args = [a.strip() for a in (m.group(2) or "").split(",")] if m.group(2) else []
    return m.group(1), [a if re.fullmatch(r"-?\d+|true|false", a) else None for a in args[1:]]


def args_match(expr_args, trace_ints) -> bool:
    return len(trace_ints) >= len(expr_args) and all(e is None or e == t for e, t in zip(expr_args, trace_ints))


def trace_args(name: str):
    """kernel name as the trace has it -> (base, [integer / bool template arguments in order]).  rocprofv3's rocpd database keeps
    the MANGLED name for kernels of an anonymous namespace (_ZN12_GLOBAL__N_1<len><name>I<template args>E...): integer arguments
    are Li<n>E / Lin<n>E, booleans Lb0E / Lb1E; type arguments (DF16b, DF16_, f) are skipped.  Demangled names are parsed too."""
    m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", name)
    if m:
        ln = int(m.group(1))
        base = name[m.end():m.end() + ln]
        rest = name[m.end() + ln:]
        ints = []
        if rest.startswith("I"):
            # the template argument list ends at the 'E' that closes it, right in front of the function's own 'E' + signature
            depth, i, tok = 1, 1, []
            while i < len(rest) and depth:
                mm = re.match(r"L([ib])(n?)(\d+)E", rest[i:])
                if mm:
                    v = mm.group(3)
                    ints.append(("true" if v == "1" else "false") if mm.group(1) == "b" else ("-" + v if mm.group(2) else v))
                    i += mm.end(); continue
                mm = re.match(r"DF16[b_]|[a-z]", rest[i:])
                if mm and rest[i] != "E":
                    i += mm.end(); continue
                if rest[i] == "E":
                    depth -= 1
                i += 1
        return base, ints
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z_]\w*)\s*(?:<(.*)>)?\s*(?:\(|$)", n)
    if not m:
        return n, []
    args, depth, cur = [], 0, ""
    for ch in (m.group(2) or ""):
        if ch == "," and depth == 0:
            args.append(cur.strip()); cur = ""
        else:
            depth += ch in "<(" ; depth -= ch in ">)"
            cur += ch
    if cur.strip():
        args.append(cur.strip())
    return m.group(1), [a for a in args if re.fullmatch(r"-?\d+|true|false", a)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("bench_json")
    ap.add_argument("db")
    ap.add_argument("--out", required=True)
    ap.add_argument("--note", default="")
    a = ap.parse_args()
    line = [l for l in Path(a.bench_json).read_text().splitlines() if l.startswith("{")][-1]
    bench = json.loads(line)
    labels = bench["labels"]
    c = sqlite3.connect(a.db)
    rows = c.execute("select name, start, end, grid_x*grid_y*grid_z/(workgroup_x*workgroup_y*workgroup_z) from kernels order by start").fetchall()
    # the replayed region: up to the longest gap in the second half of the trace (the host-side sync + set-up of the probe pass)
    half = len(rows) // 2
    gaps = [(rows[i + 1][1] - rows[i][2], i) for i in range(half, len(rows) - 1)]
    cut = max(gaps)[1] + 1 if gaps else len(rows)
    # ... and from the first launch behind the warm-up: the longest gap of the FIRST half (graph capture)
    gaps0 = [(rows[i + 1][1] - rows[i][2], i) for i in range(0, half)]
    beg = max(gaps0)[1] + 1 if gaps0 else 0
    if cut - beg < len(rows) // 2:                                 # (heuristic failed: average over everything -- the replays are > 95 % of it)
        beg, cut = 0, len(rows)
    region = rows[beg:cut]
    by = defaultdict(list)
    for name, s, e, wgs in region:
        b, ints = trace_args(name)
        by[(b, int(wgs))].append((ints, (e - s) / 1e3))
    out = {}
    for lab, rec in labels.items():
        b, ints = base_and_ints(rec["kernel"])
        cand = by.get((b, int(rec["workgroups"])), [])
        ds = [d for ti, d in cand if args_match(ints, ti)]
        out[lab] = dict(kernel=rec["kernel"], workgroups=rec["workgroups"], launches_in_trace=len(ds),
                        in_graph_avg_us=round(sum(ds) / len(ds), 2) if ds else None, probe_us=rec["probe_us"],
                        launches_per_iter=rec["launches_per_iter"], gflop=rec["gflop"])
    cfg = bench["config"]
    res = dict(config=[cfg["global_batch"] // bench["n_gpus"], cfg["img_size"], cfg["n_critic"], bench["dtype"],
                       "unet" if "U-Net" in cfg["workload"] else "simple"],
               source=f"rocprofv3 --kernel-trace of `python bench.py --no-cpu-baseline --no-also`; launches {beg}..{cut} of {len(rows)} "
                      f"(the graph replays: between the capture and the probe pass). {a.note}".strip(),
               bench_ms_per_step_under_profiler=bench["ms_per_step"], labels=out)
    Path(a.out).write_text(json.dumps(res, indent=1))
    miss = [k for k, v in out.items() if not v["in_graph_avg_us"]]
    print(f"wrote {a.out}: {len(out) - len(miss)} of {len(out)} labels matched" + (f"; unmatched: {miss}" if miss else ""))
    tot = sorted(((v["in_graph_avg_us"] or 0) * v["launches_per_iter"], k) for k, v in out.items())[::-1][:8]
    for t, k in tot:
        v = out[k]
        print(f"  {k:26s} {v['launches_per_iter']:4.1f}/iter  in-graph {v['in_graph_avg_us']} us  probe {v['probe_us']} us")


if __name__ == "__main__":
    main()
