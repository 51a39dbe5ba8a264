#!/bin/bash
# memory-path PMC passes for ONE conv configuration (run on the GPU box):  tools/archive/pmc_conv.sh fwd 768 8 128 256
# Each pass is its own rocprofv3 run with --kernel-trace only (no other trace domains), as the pool requires, and is
# bounded by its own timeout.  (A pass with the TA_* stall counters + TCP_READ_TAGCONFLICT_STALL_CYCLES aborted inside
# rocprofv3 and hung the run on this image: they are left out.  What is established about the cause: `rocprofv3 -L` lists
# all of them for gfx950 (TA_ADDR_/TA_DATA_STALLED_BY_TC_CYCLES, TA_TA_BUSY: block TA, 9 instances x 4 SEs x 8 XCCs;
# TCP_READ_TAGCONFLICT_STALL_CYCLES: block TCP), so they are not unknown names; the pass mixed two blocks whose per-pass
# slot counts MI355X_MICROARCH.md does not give (it lists SQ 8, TCC 4, GRBM 2) and its log was not kept, so an
# oversubscribed TA/TCP group is the likely cause but not a proven one.  The aborted pass hung until the timeout -- on this
# pool a hang that outlives gpurun's limit is a strike -- so it was NOT re-run to find out; the question the counters were
# meant to answer (is the gather path stalled by L1 tag conflicts?) was answered differently, by tools/archive/gather_probe.py.)
set -e; set -o pipefail
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_conv
mkdir -p $OUT
i=0
for set in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_BUSY_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o r -- python $GRAFT_REPO_ROOT/tools/conv_bench.py "$@" bf16 5 > $OUT/p$i.log 2>&1
done
ls -R $OUT | head -40
