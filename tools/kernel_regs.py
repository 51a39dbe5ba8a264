"""VGPRs / AGPRs / occupancy / scratch of every kernel of a csrc file, from hipcc -Rpass-analysis=kernel-resource-usage.
usage: python tools/kernel_regs.py norm [filter]"""
import re, subprocess, sys
from pathlib import Path
src = Path(__file__).resolve().parent.parent / "gan-calibrated-semi-supervised-learning_amd" / "csrc" / (sys.argv[1] + ".hip")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-c", str(src), "-o", "/tmp/_regs.o",
                      "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
name, d = None, {}
keys = {"VGPRs": "V", "AGPRs": "A", "Occupancy [waves/SIMD]": "occ", "ScratchSize [bytes/lane]": "scr", "LDS Size [bytes/block]": "lds"}
for l in out.splitlines():
    m = re.search(r"Function Name: (\S+)", l)
    if m:
        name, d = m.group(1), {}
    for k, s in keys.items():
        m = re.search(r"remark:\s+" + re.escape(k) + r": (\d+)", l)
        if m:
            d[s] = int(m.group(1))
    if "LDS Size" in l and name and flt in name:
        print(f"{name[:80]:80s} " + " ".join(f"{s}={d.get(s)}" for s in keys.values()))
