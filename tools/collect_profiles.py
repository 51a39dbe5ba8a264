#!/usr/bin/env python3
"""Copy the artefacts of one tools/final_run.sh call (gpurun_out/<run>/) into profiles/round4_* (the committed, judged copies).
usage: python tools/collect_profiles.py <run>          (refreshes the measured part of round4_interference.txt in place: its
header and conclusion are prose and stay)"""
import glob
import os
import re
import shutil
import sys
from pathlib import Path

R = Path(__file__).resolve().parent.parent
run = R / "gpurun_out" / sys.argv[1]
P = R / "profiles"
cp = lambda a, b: (shutil.copyfile(run / a, P / b), print(f"{a} -> profiles/{b}"))
cp("bench.json", "round4_bench.json")
cp("bench_dp1.json", "round4_bench_dp_one_rank.json")
for sub, tag in (("prof", ""), ("prof_x3", "x3_")):
    cp(f"{sub}/bench_under_rocprof.json", f"round4_{tag}bench_under_rocprof.json")
    cp(f"{sub}/kernel_summary.txt", f"round4_{tag}kernel_summary.txt")
    cp(f"{sub}/kernel_stats.csv", f"round4_{tag}kernel_stats.csv")
    cp(f"{sub}/label_durations.json", f"round4_{tag}label_durations.json")
    cp(f"{sub}/timeline.txt", f"round4_{tag}timeline_two_stream.txt")
cp("pmc/round4_pmc_dominant.json", "round4_pmc_dominant.json")
for old in glob.glob(str(P / "round4_pmc_*_pass*.csv")):
    os.remove(old)
for d in sorted((run / "pmc").iterdir()):
    if not d.is_dir():
        continue
    tag = re.sub(r"[^A-Za-z0-9]+", "_", d.name.replace(".", "")).strip("_")
    for i in range(1, 6):
        c = sorted(glob.glob(str(d / f"p{i}" / "**" / "*counter_collection.csv"), recursive=True))
        if c:
            shutil.copyfile(c[0], P / f"round4_pmc_{tag}_pass{i}.csv")
print("pmc passes:", len(glob.glob(str(P / "round4_pmc_*_pass*.csv"))))
# interference: measured tables between the header and the '# Same-box A/B' block
txt = (P / "round4_interference.txt").read_text()
head = txt[:txt.index("iterations: critic-only")]
tail = txt[txt.index("# Same-box A/B"):]
body = (run / "intf" / "interference_trace.txt").read_text().rstrip() + "\n\n" + (run / "intf" / "interference_pmc.txt").read_text().rstrip() + "\n\n"
(P / "round4_interference.txt").write_text(head + body + tail)
print("profiles/round4_interference.txt refreshed")
