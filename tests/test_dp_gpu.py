"""Data-parallel engine on the GPU: 2 ranks x (B/2) samples, gradients averaged through dist.GradAverager between
the engine's compute and update segments, must reproduce the single-process full-batch iteration.  The two ranks share
cuda:0 and talk over gloo (RCCL refuses two ranks on one device); on a multi-GPU node the same code runs with
backend "nccl" (bench.py --gpus N)."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
PKG = "gan-calibrated-semi-supervised-learning_amd"


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(rank, world, port, out_dir):
    import importlib
    sys.path.insert(0, str(ROOT))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), GCSSL_DIST_BACKEND="gloo", GCSSL_SINGLE_DEVICE="1")
    dist_mod = importlib.import_module(PKG + ".dist")
    synth = importlib.import_module(PKG + ".synth")
    engine = importlib.import_module(PKG + ".engine")
    r, w, local = dist_mod.init_from_env()
    T = torch.from_numpy
    seed, B, S, c = 21, 4, 32, 2
    g = {k: T(v) for k, v in synth.generator_state(seed).items()}
    d = {k: T(v) for k, v in synth.discriminator_state(seed).items()}
    inp = synth.step_inputs(seed, B, S, c, tag="dpgpu")
    sh = lambda a: dist_mod.shard(T(a), rank, world).contiguous().cuda()
    eng = engine.StepEngine(g, d, batch=B // world, size=S, n_critic=c, dtype="fp32", device="cuda:0",
                            allreduce=dist_mod.GradAverager() if world > 1 else None)
    refined = [sh(x) for x in inp["refined"]]
    eng.run_iteration(sh(inp["pred"]), sh(inp["gt"]), sh(inp["delta_true"]), sh(inp["pred_box"]),
                      lambda delta, k: refined[k], alphas=[sh(a).view(-1) for a in inp["alpha"]],
                      masks=[[sh(m) for m in ms] for ms in inp["masks"]])
    torch.cuda.synchronize()
    if rank == 0:
        np.savez(os.path.join(out_dir, f"w{world}.npz"), D=eng.D.p.cpu().numpy(), G=eng.G.p.cpu().numpy(),
                 u=eng.u[2].cpu().numpy())
    if world > 1:
        # replicas stay identical without ever exchanging weights
        mine = eng.D.p.cpu()
        other = [torch.zeros_like(mine) for _ in range(world)]
        torch.distributed.all_gather(other, mine)
        assert float((other[0] - other[1]).abs().max()) == 0.0
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def test_two_rank_iteration_matches_single_process(tmp_path):
    mp.spawn(_run, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    mp.spawn(_run, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    a, b = np.load(tmp_path / "w2.npz"), np.load(tmp_path / "w1.npz")
    lr = 2e-4
    for k, steps in (("D", 2), ("G", 1)):
        diff = np.abs(a[k] - b[k])
        # averaged shard gradients == full-batch gradient up to fp32 summation order; Adam's ~lr*sign(g) first steps
        # turn the rare sign flip of a ~0 gradient into a 2*lr jump: allow 1 % of those, bound the rest tightly
        assert (diff > 0.05 * lr * steps).mean() < 0.01, k
        assert diff.max() <= 2.2 * lr * steps, k
    assert np.abs(a["u"] - b["u"]).max() < 1e-5


def _run_graphed(rank, world, port, gtype):
    """The graphed data-parallel path (all-reduces started early and finished behind independent launches) against the
    eager one on the same ranks: same device-side mask / alpha draws, same collectives, so the weights must agree."""
    import importlib
    sys.path.insert(0, str(ROOT))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), GCSSL_DIST_BACKEND="gloo", GCSSL_SINGLE_DEVICE="1")
    dist_mod = importlib.import_module(PKG + ".dist")
    synth = importlib.import_module(PKG + ".synth")
    engine = importlib.import_module(PKG + ".engine")
    dist_mod.init_from_env()
    T = torch.from_numpy
    seed, B, S, c = 23, 8, 32, 2
    gtype = gtype or "unet"
    g = {k: T(v) for k, v in (synth.simple_generator_state(seed) if gtype == "simple" else synth.generator_state(seed)).items()}
    d = {k: T(v) for k, v in synth.discriminator_state(seed).items()}
    inp = synth.step_inputs(seed, B, S, c, tag="dpgraph", generator_type=gtype)
    sh = lambda a: dist_mod.shard(T(a), rank, world).contiguous().cuda()
    refined = [sh(x) for x in inp["refined"]]
    call = (sh(inp["pred"]), sh(inp["gt"]), sh(inp["delta_true"]), sh(inp["pred_box"]), lambda delta, k: refined[k])
    mk = lambda: engine.StepEngine(g, d, batch=B // world, size=S, n_critic=c, dtype="fp32", device="cuda:0",
                                   seed=77 + rank, allreduce=dist_mod.GradAverager(), keep_clipped_grads=False,
                                   generator_type=gtype)
    eager, graphed = mk(), mk()
    for _ in range(2):
        eager.run_iteration(*call)
    gi = engine.GraphedIteration(graphed, *call)
    for _ in range(2):
        gi.replay()
    torch.cuda.synchronize()
    lr = 2e-4
    for name, a, b, steps in (("D", eager.D.p, graphed.D.p, 4), ("G", eager.G.p, graphed.G.p, 2)):
        diff = (a - b).abs()
        assert float((diff > 0.05 * lr * steps).float().mean()) < 0.02, (name, float((diff > 0.05 * lr * steps).float().mean()))
        assert float(diff.max()) <= 2.2 * lr * steps, name
    assert float(graphed.D.state[0]) == 4.0 and float(graphed.G.state[0]) == 2.0
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("gtype", ["unet", "simple"])
def test_graphed_overlapped_allreduce_matches_eager(gtype):
    mp.spawn(_run_graphed, args=(2, _free_port(), gtype), nprocs=2, join=True)


def _run_graphed_bench_shape(rank, world, port, pipeline, compress):
    """the data-parallel graph schedule at the BENCH shape -- 256 samples per rank, bf16, n_critic = 2 -- on two
    ranks (gloo, sharing cuda:0): `pipeline` 0 = the default single-communicator dp_branch schedule, 1 = the two-communicator
    pipelined one (GCSSL_DP_PIPELINE=1, opt-in since round 4).  Three replays against three eager iterations of a twin engine on
    the same ranks: finite, the optimiser step counts, no weight further apart than Adam's steps allow, and -- the property
    data parallelism rests on -- the replicas of BOTH ranks bit-identical after every exchange form."""
    import importlib
    sys.path.insert(0, str(ROOT))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), GCSSL_DIST_BACKEND="gloo", GCSSL_SINGLE_DEVICE="1", GCSSL_DP_PIPELINE=str(pipeline))
    dist_mod = importlib.import_module(PKG + ".dist")
    synth = importlib.import_module(PKG + ".synth")
    engine = importlib.import_module(PKG + ".engine")
    dist_mod.init_from_env()
    T = torch.from_numpy
    seed, Bt, S, c = 31, 512, 32, 2
    g = {k: T(v) for k, v in synth.generator_state(seed).items()}
    d = {k: T(v) for k, v in synth.discriminator_state(seed).items()}
    inp = synth.step_inputs(seed, Bt, S, c, tag="dpbench")
    sh = lambda a: dist_mod.shard(T(a), rank, world).contiguous().cuda()
    refined = [sh(x) for x in inp["refined"]]
    call = (sh(inp["pred"]), sh(inp["gt"]), sh(inp["delta_true"]), sh(inp["pred_box"]), lambda delta, k: refined[k])
    mk = lambda: engine.StepEngine(g, d, batch=Bt // world, size=S, n_critic=c, dtype="bf16", device="cuda:0", seed=77 + rank,
                                   allreduce=dist_mod.GradAverager(compress=compress), keep_clipped_grads=False)
    eager, graphed = mk(), mk()
    gi = engine.GraphedIteration(graphed, *call)
    assert not gi.fused_update and gi.dp_pipeline == bool(pipeline)
    for _ in range(3):
        eager.run_iteration(*call)
        gi.replay()
    torch.cuda.synchronize()
    lr = 2e-4
    for name, a, b, steps in (("D", eager.D.p, graphed.D.p, 6), ("G", eager.G.p, graphed.G.p, 3)):
        assert bool(torch.isfinite(a).all()) and bool(torch.isfinite(b).all()), name
        assert float((a - b).abs().max()) <= 2.2 * lr * steps, (name, float((a - b).abs().max()))
    assert float(graphed.D.state[0]) == 6.0 and float(graphed.G.state[0]) == 3.0
    assert graphed.saturations() == {"critic": 0, "generator": 0}
    for eng in (eager, graphed):                                   # replicas stay identical without ever exchanging weights
        for flat in (eng.D.p, eng.G.p):
            mine = flat.cpu()
            both = [torch.zeros_like(mine) for _ in range(world)]
            torch.distributed.all_gather(both, mine)
            assert torch.equal(both[0], both[1])
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("pipeline,compress", [(0, None), (1, None), (0, "bf16")])
def test_graphed_dp_schedule_at_bench_shape(pipeline, compress):
    mp.spawn(_run_graphed_bench_shape, args=(2, _free_port(), pipeline, compress), nprocs=2, join=True)


def _run_rccl_one_rank(rank, world, port):
    """The data-parallel schedule on the REAL backend (nccl == RCCL) with one rank: process group, RCCL communicator, async
    all-reduces between the captured graph segments (GCSSL_FORCE_DP=1 makes a world of 1 take that path) -- against the
    single-GPU one-graph form on the same device draws."""
    import importlib
    sys.path.insert(0, str(ROOT))
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      GCSSL_FORCE_DP="1")
    os.environ.pop("GCSSL_DIST_BACKEND", None)
    os.environ.pop("GCSSL_SINGLE_DEVICE", None)
    dist_mod = importlib.import_module(PKG + ".dist")
    synth = importlib.import_module(PKG + ".synth")
    engine = importlib.import_module(PKG + ".engine")
    dist_mod.init_from_env()
    assert torch.distributed.is_initialized() and torch.distributed.get_backend() == "nccl"
    assert torch.distributed.get_world_size() == 1                       # bench.py's `rccl_ranks`
    T = torch.from_numpy
    seed, B, S, c = 29, 16, 32, 2
    g = {k: T(v) for k, v in synth.generator_state(seed).items()}
    d = {k: T(v) for k, v in synth.discriminator_state(seed).items()}
    inp = synth.step_inputs(seed, B, S, c, tag="rccl1")
    refined = [T(x).cuda() for x in inp["refined"]]
    call = (T(inp["pred"]).cuda(), T(inp["gt"]).cuda(), T(inp["delta_true"]).cuda(), T(inp["pred_box"]).cuda(),
            lambda delta, k: refined[k])
    avg = dist_mod.GradAverager()
    assert avg.world == 1
    mk = lambda ar: engine.StepEngine(g, d, batch=B, size=S, n_critic=c, dtype="fp32", device="cuda:0", seed=91,
                                      allreduce=ar, keep_clipped_grads=False)
    one, dp = mk(None), mk(avg)
    g_one, g_dp = engine.GraphedIteration(one, *call), engine.GraphedIteration(dp, *call)
    assert g_one.fused_update and not g_dp.fused_update                   # one graph per iteration vs the DP segments
    assert g_dp.dp_pipeline == (os.environ.get("GCSSL_DP_PIPELINE", "1") != "0")
    for _ in range(2):
        g_one.replay()
        g_dp.replay()
    torch.cuda.synchronize()
    lr = 2e-4
    for name, a, b, steps in (("D", one.D.p, dp.D.p, 4), ("G", one.G.p, dp.G.p, 2)):
        diff = (a - b).abs()
        assert bool(torch.isfinite(b).all())
        assert float((diff > 0.05 * lr * steps).float().mean()) < 0.02, (name, float((diff > 0.05 * lr * steps).float().mean()))
        assert float(diff.max()) <= 2.2 * lr * steps, name
    assert float(dp.D.state[0]) == 4.0 and float(dp.G.state[0]) == 2.0
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("pipeline", ["1", "0"])
def test_rccl_backend_single_rank_graph_segments(pipeline, monkeypatch):
    """the nccl (RCCL) path under -m gpu -- a fresh child process, one rank.  pipeline 1: the generator's
    chain with its own communicator on the second stream (the default); 0: the first round-3 schedule."""
    monkeypatch.setenv("GCSSL_DP_PIPELINE", pipeline)          # (the spawned child inherits the environment)
    mp.spawn(_run_rccl_one_rank, args=(1, _free_port()), nprocs=1, join=True)
