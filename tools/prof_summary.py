"""Summarise a rocprofv3 rocpd database (kernel trace) per kernel name: ms/iteration, launches/iteration, avg us.
usage: python tools/prof_summary.py gpurun_out/prof/r_results.db [--csv out.csv]"""
import re
import sqlite3
import sys


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(end-start), avg(end-start), max(grid_x*grid_y*grid_z/(workgroup_x*workgroup_y*workgroup_z))"
                     " from kernels group by name order by 3 desc").fetchall()
    # one kernel name serves several layers: also report its LARGEST launch class (durations within 25 % of the longest),
    # which is what a per-layer figure such as bench.py's roofline.avg_us of the dominant launch has to agree with
    top = {}
    for name, dur in c.execute("select name, end-start from kernels").fetchall():
        top.setdefault(name, []).append(dur)
    for name, ds in top.items():
        mx = max(ds)
        cls = [d for d in ds if d >= 0.75 * mx]
        top[name] = (mx / 1e3, sum(cls) / len(cls) / 1e3, len(cls))
    n_it = [r[1] for r in rows if "adam_kernel" in r[0]][0] / 3          # three optimiser updates per iteration
    tot = sum(r[2] for r in rows)
    print(f"iterations {n_it:.0f}  kernel-busy {tot / 1e6 / n_it:.3f} ms/iter  launches/iter {sum(r[1] for r in rows) / n_it:.1f}")
    out = []
    for n, cnt, s, a, g in rows:
        raw = n
        n = n.replace("(anonymous namespace)::", "").replace("void ", "")
        n = re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", n)
        m = re.match(r"_ZN12_GLOBAL__N_1\d+(\w+?)I", n)
        if m:
            n = m.group(1) + "<" + n[m.end() - 1:][:40] + ">"
        out.append((s / 1e6 / n_it, cnt / n_it, a / 1e3, g, n[:100], top[raw][1], top[raw][2] / n_it))
    for r in out[:60]:
        print(f"{r[0]:7.3f} ms/it {r[1]:6.1f}/it {r[2]:8.1f} us  wg{r[3]:6d} {r[4]}")
    if "--csv" in sys.argv:
        with open(sys.argv[sys.argv.index("--csv") + 1], "w") as f:
            f.write("kernel,ms_per_iter,launches_per_iter,avg_us,max_workgroups,largest_class_avg_us,largest_class_launches_per_iter\n")
            for r in out:
                f.write(f"\"{r[4]}\",{r[0]:.4f},{r[1]:.2f},{r[2]:.2f},{r[3]},{r[5]:.2f},{r[6]:.2f}\n")


if __name__ == "__main__":
    main()
