"""GPU time of gcssl_conv4x4s2_dgrad_act_bwd at the critic's c2 shape (N x 16 x 16 x 64 <- N x 8 x 8 x 128), 40 launches per
graph replay.  usage: python tools/archive/actb_bench.py [N] [eager launches instead of the graph, for rocprofv3 --pmc passes]"""
import importlib, sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
ops = importlib.import_module("gan-calibrated-semi-supervised-learning_amd.ops")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 768
dt = torch.bfloat16
dy = torch.randn(N, 8, 8, 128, device="cuda").to(dt); a = torch.randn(N, 16, 16, 64, device="cuda").to(dt)
w = torch.randn(128, 64, 4, 4, device="cuda") * 0.05
wf = torch.empty(128, 16, 64, device="cuda", dtype=dt); wt = torch.empty(64, 16, 128, device="cuda", dtype=dt)
ops.prep_conv_weight(w, wf, wt, 128, 64, 64, ops.code(wf))
dz = torch.empty(N, 16, 16, 64, device="cuda", dtype=dt)
gs = torch.ones(3, device="cuda"); bias = torch.zeros(64, device="cuda")
rep = torch.zeros(4, 128, device="cuda")
def run(): ops.conv_dgrad_act_bwd(dy, wt, a, dz, 64, 128, gscale=gs, group_n=N // 3, bias=bias, dbias=rep[0, :64], cdot=rep[0, 64:67], nrep=4, rep_stride=128)
for _ in range(3): run()
torch.cuda.synchronize()
if len(sys.argv) > 2:
    for _ in range(int(sys.argv[2])): run()
    torch.cuda.synchronize()
    print(f"N={N}: {sys.argv[2]} eager launches  {ops.last_kernel()}")
    sys.exit(0)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(40): run()
g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): g.replay()
e1.record(); torch.cuda.synchronize()
print(f"N={N}: {e0.elapsed_time(e1) / 400 * 1e3:.1f} us per launch  {ops.last_kernel()}")
