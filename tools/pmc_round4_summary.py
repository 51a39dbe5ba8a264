#!/usr/bin/env python3
"""Fold tools/pmc_round4.sh's passes into one JSON: per engine label the HBM bytes of one launch (FETCH_SIZE doubled per
MI355X_MICROARCH.md 'HBM', + WRITE_SIZE; both KiB), its duration under the profiler, and the SQ counters with the MFMA-busy
fraction (SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CYCLES per CU ...) is left to the reader: raw sums are stored).
usage: python tools/pmc_round4_summary.py gpurun_out/pmc_r4 bf16 round4_pmc_dominant.json tools/pmc_round4.sh"""
import csv, glob, json, os, sys
from collections import defaultdict

out_dir, dt = sys.argv[1], sys.argv[2]
out_name = sys.argv[3] if len(sys.argv) > 3 else "round4_pmc_dominant.json"
script = sys.argv[4] if len(sys.argv) > 4 else "tools/pmc_round4.sh"
CASES = {"D.c2-4.wgrad": "conv_wgrad_dma_batch", "D.c2.fwd[n=768]": "conv_dma", "D.c3.fwd[n=768]": "conv_dma", "D.c4.fwd[n=768]": "conv_dma", "D.c2.wgrad": "conv_wgrad",
         "D.c3.wgrad": "conv_wgrad", "D.c4.wgrad": "conv_wgrad", "D.c2.dgrad": "dgrad_img", "D.c3.dgrad": "conv_dma", "D.c4.dgrad": "conv_dma", "D.c1.fwd[n=768]": "conv_",
         "D.c1.gp_dgrad": "conv_", "G.up4.fwd[n=768]": "convt_in_relu",
         # the split-precision mode's leading launches (fp16x3: recorded under their own tags) and the re-crop stage
         "x3.D.c2.fwd[n=768]": "conv_fwd_kernel", "x3.D.c3.dgrad": "conv_dgrad_kernel", "x3.D.c2.wgrad": "conv_wgrad_kernel",
         "recrop[B=256,1280x720]": "recrop_kernel"}
labels = {}
for tag, flt in CASES.items():
    d = os.path.join(out_dir, tag.translate(str.maketrans("[]=,", "____")))
    if not os.path.isdir(d):
        continue
    rec = {"kernel_filter": flt, "counters_avg_per_launch": {}}
    for i in (1, 2, 3, 4, 5):
        f = glob.glob(f"{d}/p{i}/**/*counter_collection.csv", recursive=True)
        if not f:
            continue
        acc = defaultdict(list)
        kname = None
        for r in csv.DictReader(open(f[0])):
            if flt in r["Kernel_Name"] and "prep" not in r["Kernel_Name"] and "reduce" not in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"])); kname = r["Kernel_Name"]
        for k, v in acc.items():
            v = v[len(v) // 2:]                                   # the timed (warm) launches
            rec["counters_avg_per_launch"][k] = sum(v) / len(v)
        rec["kernel"] = (kname or "")[:160]
        kt = glob.glob(f"{d}/p{i}/**/*kernel_trace.csv", recursive=True)
        if kt and i == 1:
            ds = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(kt[0]))
                  if flt in r["Kernel_Name"] and "prep" not in r["Kernel_Name"] and "reduce" not in r["Kernel_Name"]]
            ds = ds[len(ds) // 2:]
            rec["avg_us_under_pmc"] = sum(ds) / max(len(ds), 1)
    c = rec["counters_avg_per_launch"]
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        rec["hbm_fetch_bytes_corrected_x2"] = c["FETCH_SIZE"] * 1024 * 2
        rec["hbm_write_bytes"] = c["WRITE_SIZE"] * 1024
        rec["hbm_bytes_per_launch"] = rec["hbm_fetch_bytes_corrected_x2"] + rec["hbm_write_bytes"]
    if "SQ_ACTIVE_INST_VALU" in c and "SQ_WAVE_CYCLES" in c:
        # both count quad-cycles of wave time: the share of its waves' lifetime a kernel spends issuing VALU / LDS instructions
        rec["valu_issue_frac_of_wave_time"] = c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"]
        rec["lds_issue_frac_of_wave_time"] = c.get("SQ_ACTIVE_INST_LDS", 0.0) / c["SQ_WAVE_CYCLES"]
    if "SQ_WAIT_INST_ANY" in c and "SQ_WAVE_CYCLES" in c:
        rec["waiting_frac_of_wave_time"] = c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
        # one 32x32x16 MFMA = 32 busy cycles (8 passes x 4); GRBM_GUI_ACTIVE counts all 8 XCDs
        rec["mfma_busy_frac_of_simd_time"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * c["GRBM_GUI_ACTIVE"] / 8.0)
    labels[tag] = rec
res = dict(config=[256, 32, 2, dt, "unet"], x3_tags="labels prefixed x3. are the fp16x3 mode's launches of the same shapes (fp32 tensors)",
           command=script + " (rocprofv3 --kernel-trace --pmc <set>, separate passes, tools/conv_bench.py / tools/convt_bench.py "
                   "launching the label's shape stand-alone)",
           note="FETCH_SIZE / WRITE_SIZE in KiB; FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md). The stand-alone "
                "launch re-reads the same operands each repetition, so Infinity-Cache hits are counted as the guide says; wgrad bytes "
                "include the fp32 slabs.",
           labels=labels)
json.dump(res, open(os.path.join(out_dir, out_name), "w"), indent=1)
for k, v in labels.items():
    print(k, {kk: (round(vv, 1) if isinstance(vv, float) else vv) for kk, vv in v.items() if kk not in ("counters_avg_per_launch", "kernel", "kernel_filter")})
