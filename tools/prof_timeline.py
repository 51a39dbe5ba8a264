"""Timeline of ONE steady-state iteration from a rocprofv3 rocpd database (kernel trace): every launch in start order with
its start offset, duration, queue, and how many other kernels overlap it; then the union-busy time, the time with exactly
one / two+ kernels resident, and the idle time.
usage: python tools/prof_timeline.py r_results.db [iteration index from the end, default 20] [marker kernel, default eiou_kernel]
       [marker occurrences per iteration, default 1]"""
import re
import sqlite3
import sys


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    n = re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", n)
    m = re.match(r"_ZN12_GLOBAL__N_1\d+(\w+?)I", n)
    if m:
        n = m.group(1) + "<" + n[m.end() - 1:][:44] + ">"
    return n[:70]


def main():
    c = sqlite3.connect(sys.argv[1])
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    cols = [r[1] for r in c.execute("pragma table_info(kernels)").fetchall()]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    rows = c.execute(f"select name, start, end, {qcol or 0} from kernels order by start").fetchall()
    marker = sys.argv[3] if len(sys.argv) > 3 else "eiou_kernel"               # once per iteration (generator branch)
    every = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    marks = [i for i, r in enumerate(rows) if marker in r[0]][::every]
    i0, i1 = marks[-back - 1], marks[-back]
    # an iteration's launches: between two consecutive markers (the window is shifted, but it is one period)
    it = rows[i0:i1]
    t0 = it[0][1]
    ev = []
    for n, s, e, q in it:
        ev.append((s, 1)); ev.append((e, -1))
    ev.sort()
    busy1 = busy2 = 0; idle = 0; depth = 0; last = ev[0][0]
    for t, d in ev:
        if depth == 1: busy1 += t - last
        elif depth >= 2: busy2 += t - last
        else: idle += t - last
        depth += d; last = t
    period = it[-1][2] - t0
    print(f"launches {len(it)}  period {period / 1e3:.1f} us  one kernel resident {busy1 / 1e3:.1f} us  two or more {busy2 / 1e3:.1f} us  idle {idle / 1e3:.1f} us")
    qs = sorted({r[3] for r in it})
    for n, s, e, q in it:
        ov = sum(1 for m, s2, e2, q2 in it if s2 < e and e2 > s) - 1
        print(f"{(s - t0) / 1e3:8.1f} us  +{(e - s) / 1e3:6.1f}  q{qs.index(q)}  ov{ov}  {short(n)}")


if __name__ == "__main__":
    main()
