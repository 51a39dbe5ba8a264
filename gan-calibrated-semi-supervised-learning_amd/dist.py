"""Data parallelism for the step engine: one process per GPU, torch.distributed (backend "nccl" == RCCL over xGMI on
ROCm; "gloo" for the CPU tests), pure data parallel with ONE exchange per optimiser step.

The reference is single-process (cgan/cgan_train_enhanced.py:171); this is a new capability.  Every per-sample
quantity of the step is independent across samples (InstanceNorm is per (n,c); GP is a per-sample norm then a batch
mean), so with equal shards the AVERAGE of the shard gradients equals the single-process gradient.  The engine keeps
each network's gradient in one flat fp32 buffer, which is therefore the all-reduce bucket: D grads 11.07 MB per
critic step, G grads 25.18 MB per iteration.  clip_grad_norm_ runs after the all-reduce on the averaged gradient,
exactly like single-process semantics (:331,:368), so the norm itself needs no collective.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
from typing import Optional, Sequence

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> tuple:
    """Initialise the default process group from RANK/WORLD_SIZE/LOCAL_RANK/MASTER_* (torchrun contract).
    Returns (rank, world_size, local_rank).  Single-process runs need no process group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # GCSSL_FORCE_DP=1: rehearse the data-parallel code path (process group, async all-reduce between the graph segments)
    # with a single rank on the real RCCL backend -- the only way to exercise it on a one-GPU box
    if (world > 1 or force_dp()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # GCSSL_DIST_BACKEND=gloo lets several ranks rehearse the DP path on ONE GPU (RCCL refuses duplicate devices)
            backend = os.environ.get("GCSSL_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    if os.environ.get("GCSSL_SINGLE_DEVICE"):        # rehearsal mode: every rank drives cuda:0
        local = 0
    return rank, world, local


def force_dp() -> bool:
    return os.environ.get("GCSSL_FORCE_DP", "0") == "1"


class GradAverager:
    """all-reduce(sum)/world of a flat gradient buffer, in place.  Callable, used as ``StepEngine(allreduce=...)``.

    compress="bf16" (or GCSSL_AR_DTYPE=bf16): the bucket crosses the wire as bf16 -- half the bytes over xGMI (the
    ring time of the 11 MB + 11 MB + 25 MB per iteration is 0.55 ms at 8 GPUs against a 1.9 ms step).  The fp32 bucket is
    rounded into a bf16 staging buffer, RCCL sums in bf16, the result is widened back over the bucket: every summand carries
    a relative rounding of 2^-9 and the ring adds log-depth more -- measured 4e-3 of the gradient's norm at world 2
    (tests/test_dist_cpu.py), below what the bf16 compute mode itself leaves on a gradient (1e-2), far above the fp32
    modes' 1e-6: a throughput option for the 16-bit modes, not for the parity modes."""

    def __init__(self, group=None, compress: Optional[str] = None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        compress = compress if compress is not None else os.environ.get("GCSSL_AR_DTYPE", "")
        if compress not in ("", "fp32", "bf16"):
            raise ValueError(f"GradAverager: compress must be 'bf16' or unset, got {compress!r}")
        self.compress = compress == "bf16"
        self._stage = {}                       # data_ptr of a bucket -> its bf16 staging buffer

    def _staged(self, flat: torch.Tensor) -> torch.Tensor:
        b = self._stage.get(flat.data_ptr())
        if b is None or b.numel() != flat.numel():
            b = self._stage[flat.data_ptr()] = torch.empty(flat.numel(), dtype=torch.bfloat16, device=flat.device)
        return b

    def __call__(self, flat: torch.Tensor) -> None:
        if self.world == 1 and not force_dp():
            return
        self.finish(self.start(flat), flat)

    # split form: the collective runs on the backend's own stream (RCCL) / thread (gloo) while the caller keeps launching
    # work that does not need the result; finish() orders the caller's stream behind it.
    def start(self, flat: torch.Tensor):
        if self.world == 1 and not force_dp():
            return None
        if self.compress:
            b = self._staged(flat)
            b.copy_(flat)                                            # fp32 -> bf16 (RNE) on the caller's stream, in front of the collective
            return (dist.all_reduce(b, op=dist.ReduceOp.SUM, group=self.group, async_op=True), b)
        return dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish(self, handle, flat: torch.Tensor, scale: bool = True) -> None:
        """scale=False: only order the stream behind the collective; the caller applies 1/world itself (the step engine
        folds it into its fused clip+Adam launch: StepEngine._grad_scale)."""
        if handle is None:
            return
        if isinstance(handle, tuple):
            work, b = handle
            work.wait()
            flat.copy_(b)                                            # the bf16 sum, widened back over the fp32 bucket
        else:
            handle.wait()
        if scale:
            flat.mul_(1.0 / self.world)


def shard(t: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """Equal split along the batch dim (dim 0); the batch must divide evenly so mean-of-means == global mean."""
    if t.shape[0] % world:
        raise ValueError(f"batch {t.shape[0]} is not divisible by world size {world}")
    n = t.shape[0] // world
    return t[rank * n:(rank + 1) * n]


def broadcast_state(tensors, src: int = 0, group=None) -> None:
    """Replicate parameters / spectral-norm u,v from rank `src` (replicas then stay identical: every rank runs the
    same number of train-mode critic forwards on the same weights)."""
    if not dist.is_initialized():
        return
    for t in tensors:
        dist.broadcast(t, src=src, group=group)


def launch_local_ranks(argv: Sequence[str], nproc: int, extra_env: Optional[dict] = None, timeout: Optional[float] = None):
    """Start `nproc` fresh child processes of `argv` (one rank each: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
    MASTER_PORT set, rendezvous on 127.0.0.1) and wait for them.  The caller must not have touched the GPU: the children
    are new processes (never an exec of this one).  Rank 0's stdout is captured and returned; the other ranks' stdout goes
    to this process's stderr.  -> (exit codes, rank 0's stdout).

    EVERY child is polled: the first rank that exits non-zero -- whichever it is, also one that dies before the rendezvous
    while rank 0 sits in init_process_group or in a collective -- gets its peers terminated at once instead of leaving them
    to the process group's own timeout (10+ minutes).  `timeout` bounds the whole launch the same way."""
    import threading
    import time
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(nproc):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nproc), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), **(extra_env or {}))
        procs.append(subprocess.Popen(list(argv), env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True))
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)   # drain rank 0's pipe
    reader.start()
    deadline = None if timeout is None else time.monotonic() + timeout
    try:
        while True:
            codes = [pr.poll() for pr in procs]
            if all(c is not None for c in codes) or any(c not in (None, 0) for c in codes):
                break
            if deadline is not None and time.monotonic() > deadline:
                break
            time.sleep(0.05)
    finally:
        for pr in procs:
            if pr.poll() is None:                     # a rank that outlives a failed / timed-out peer would hang in a collective
                pr.terminate()
        for pr in procs:
            try:
                pr.wait(timeout=10)
            except subprocess.TimeoutExpired:
                pr.kill()
                pr.wait()
        reader.join(timeout=10)
    return [pr.returncode for pr in procs], "".join(c or "" for c in chunks)
