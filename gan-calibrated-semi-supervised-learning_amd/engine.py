"""Explicit-schedule step engine: the WGAN-GP cGAN training iteration of the reference
(cgan/cgan_train_enhanced.py:304-369) as a fixed sequence of HIP kernel launches on one stream -- no autograd
tape, no host synchronisation, every buffer preallocated (so the whole iteration can be captured in a hipGraph).

MI355X-first choices:
  * the three critic forwards of a D step (real, fake, interpolated; :308,:316 and cgan/losses.py:210) run as ONE
    3B-sample batch; their different spectral-norm sigmas become a per-sample-group scale in the conv epilogue;
  * the WGAN-GP double backward (cgan/losses.py:213-220 + :330) is written out by hand (oracle/manual_step.py is
    the executable spec): first-order dgrad chain, its reverse (forward-conv chain + wgrad + InstanceNorm
    double-backward), then one batched backward of the three forwards;
  * activations are NHWC in the compute dtype, concat buffers are written in place, weight gradients go
    split-K slab -> fused reduce (+ spectral-norm rank-1 correction) straight into one flat fp32 gradient buffer
    that is also the RCCL all-reduce bucket and the fused clip+Adam operand.
"""
from __future__ import annotations

import math
import os
from typing import Callable, Dict, Optional, Sequence

import torch

from . import _lib, ops
from .gen_simple import GS_PARAM_KEYS, SimpleGenerator
from .ops import LRELU, RELU

D_IDX = (0, 2, 5, 8)
D_CH = [(6, 64), (64, 128), (128, 256), (256, 512)]            # (Cin, Cout) of the SN convs
G_DOWN = [(3, 64), (64, 128), (128, 256), (256, 512)]
G_UP = [(512, 256), (512, 128), (256, 64), (128, 64)]          # ConvTranspose (CinT, CoutT)
D_PARAM_KEYS = [k for i in D_IDX for k in (f"model.{i}.bias", f"model.{i}.weight_orig")] + ["model.11.weight"]
G_PARAM_KEYS = ["down1.model.0.weight", "down2.model.0.weight", "down3.model.0.weight", "down4.model.0.weight",
                "up1.model.0.weight", "up2.model.0.weight", "up3.model.0.weight", "up4.0.weight",
                "fc_delta.1.weight", "fc_delta.1.bias"]


def _pad8(c: int) -> int:
    return max(8, c)


# Side streams are shared by every engine of a process (one per device and role).  HIP deals streams to a small pool of hardware
# queues round-robin; an engine that creates fresh streams every time ends up, after a few engines in one process, with its side
# stream on the SAME hardware queue as the compute stream -- the two chains of the iteration then run one after the other.
# (Round 4: bench.py's fifth configuration, fp16x3, read 57.0k images/s as an `also` entry and 67-69k as a run of its own.)
_STREAMS: Dict[tuple, "torch.cuda.Stream"] = {}


def _side_stream(dev, role: str, priority: int = 0) -> "torch.cuda.Stream":
    if role == "gen" and priority == 0:
        priority = int(os.environ.get("GCSSL_GEN_PRIO", "0"))       # A/B knob: the generator chain's stream at another priority
    key = (str(dev), role, priority)
    if key not in _STREAMS:
        for i in range(int(os.environ.get("GCSSL_SIDE_SKIP", "0"))):  # A/B knob: which hardware queue the side stream lands on
            _STREAMS[(str(dev), f"skip{i}", priority)] = torch.cuda.Stream(device=dev, priority=priority)
        _STREAMS[key] = torch.cuda.Stream(device=dev, priority=priority)
    return _STREAMS[key]


def _algorithmic_bytes(label: str, args, es: int):
    """-> (algorithmic, stored) bytes of one conv launch.  Algorithmic is SURVEY.md 8(d)'s definition: the input read once
    + the result written once + the weights read once, every element at the compute dtype's size `es` (a wgrad's result is
    ONE weight-gradient tensor, not one slab per K split).  Stored is what the launch's own tensors occupy (fp32 pre-norm
    outputs, fp32 slabs): stored - algorithmic is traffic the design adds.  Tensors are taken from the launch's arguments."""
    ts = [a for a in args if isinstance(a, torch.Tensor)]
    algo = stored = 0
    for i, t in enumerate(ts):
        n = t.numel()
        stored += n * t.element_size()
        if label.endswith("wgrad") and t.dim() == 4 and t.dtype == torch.float32 and i == 2:
            n //= t.shape[0]                                          # slab [nsplit][Cout][16][Cin]: ONE gradient tensor
        algo += n * es
    return algo, stored


def roofline_bound(label: str) -> str:
    """Which roof bounds a conv launch, per SURVEY.md 8(d): MFMA for every layer with >= 64 input channels (D.c2-c4,
    G.down2-4, G.up1-4: 334-910 FLOP/B in a 16-bit dtype against a ridge of 312), HBM for the 8-channel first layers."""
    layer = label.split(".")[1] if "." in label else label
    return "hbm" if layer in ("c1", "down1") else "mfma"


class FlatParams:
    """Parameters of one network as views into one flat fp32 buffer (+ grad / Adam moments of the same shape)."""

    def __init__(self, sd: Dict[str, torch.Tensor], keys: Sequence[str], device):
        self.keys = list(keys)
        n = sum(sd[k].numel() for k in keys)
        self.p = torch.empty(n, device=device)
        self.g = torch.zeros(n, device=device)
        self.m = torch.zeros(n, device=device)
        self.v = torch.zeros(n, device=device)
        self.grads_zero = True                                            # g is all-zero (fresh, or zeroed by the last update)
        self.state = torch.zeros(ops.ADAM_STATE, device=device, dtype=torch.float64)   # step, sumsq, last grad norm, ... (gcssl.h)
        self.views, self.gviews, off = {}, {}, 0
        for k in keys:
            t = sd[k]
            self.views[k] = self.p[off:off + t.numel()].view(t.shape)
            self.gviews[k] = self.g[off:off + t.numel()].view(t.shape)
            self.views[k].copy_(t)
            off += t.numel()

    def moment_views(self, k: str):
        off = 0
        for kk in self.keys:
            n = self.views[kk].numel()
            if kk == k:
                return self.m[off:off + n].view(self.views[kk].shape), self.v[off:off + n].view(self.views[kk].shape)
            off += n
        raise KeyError(k)


class _NoSpectralNorm:
    """Stand-in for ops.SnState when the critic is built with spectral_norm=False (cgan/models.py:228-238, config.yaml
    `spectral_norm: false`): sigma = 1 for every layer and forward, nothing to iterate; only the fill that rides on the
    closing launch of a power-iteration chain is left to do."""

    def __init__(self, n, nslots, device):
        self.isig = torch.ones(n, nslots, device=device)
        self.sigma = torch.ones(n, nslots, device=device)
        self.u_hist = self.v_hist = None

    def iterate(self, slot, iterate=True, zero=None, defer_finish=False):
        if zero is not None:
            zero.zero_()


class GFwd:
    """Forward-pass buffers of the U-Net generator for a batch of n samples (activations, pre-norm tensors with their
    split-K slab room, statistics, dropout masks, head)."""
    FIELDS = ("cat3", "cat2", "cat1", "d4", "u4", "pooled", "poolsum", "ucnt", "traw", "delta", "x8")
    LISTS = ("zd", "zu", "dmean", "drstd", "umean", "urstd", "masks")

    def group(self, g: int, b: int) -> "GFwd":
        """views of samples [g*b, (g+1)*b) -- the buffers one generator call of the iteration reads and writes"""
        sl, v = slice(g * b, (g + 1) * b), GFwd()
        v.n, v.maskbuf = b, None
        for f in self.FIELDS:
            setattr(v, f, getattr(self, f)[sl])
        for f in self.LISTS:
            setattr(v, f, [None if t is None else t[sl] for t in getattr(self, f)])
        return v


def conv_flops(n: int, hi: int, cin: int, cout: int) -> float:
    """Algorithmic FLOPs (2*MAC) of one k4 s2 p1 conv pass (forward, dgrad or wgrad) over n samples of hi x hi input."""
    return 2.0 * n * (hi // 2) * (hi // 2) * cout * 16 * cin


class StepEngine:
    """Holds G and D (weights, Adam state, spectral-norm u/v) on one GPU and runs reference-ordered iterations.

    dtype: "fp32" (exact fp32 MFMA; the parity mode), "bf16" or "fp16" (16-bit MFMA operands, fp32 accumulate: the
    throughput modes).  fp16 has 3 more mantissa bits than bf16 and 5 exponent bits: the backward passes run on gradients
    multiplied by a static loss scale (loss_scale_d / loss_scale_g, powers of two: every backward quantity is linear in its
    seed) and the fused clip+Adam launch divides it out again, so small gradients stay in fp16's normal range; losses,
    statistics, weight gradients and optimiser state are fp32 (BASELINE configs[3]: "fp16 with fp32 loss accum").
    ``allreduce(flat_grad)`` (optional) is called on the flat D / G gradient right before clip+Adam: the data-parallel
    hook (dist.py).  ``refine_fn(delta, k)`` stands in for the host re-crop stage ``get_refined_patch_batch``
    (cgan/cgan_train_enhanced.py:37-137): it must return a (B,3,S,S) fp32 NCHW tensor with no dependence path
    back into the engine's buffers.
    """

    def __init__(self, sd_g: Dict[str, torch.Tensor], sd_d: Dict[str, torch.Tensor], batch: int, size: int,
                 n_critic: int = 2, dtype="bf16", device="cuda", lr: float = 2e-4, betas=(0.5, 0.999),
                 delta_scale: float = 0.3, lambda_gp: float = 1.0, lambda_iou: float = 1.0, seed: int = 42,
                 allreduce: Optional[Callable[[torch.Tensor], None]] = None, keep_clipped_grads: bool = True,
                 overlap: int = 0, generator_type: str = "unet", spectral_norm: bool = True):
        if generator_type not in ("unet", "simple"):
            raise ValueError(f"generator_type must be 'unet' or 'simple' (cgan/cgan_train_enhanced.py:26-31), got {generator_type!r}")
        self.generator_type = generator_type
        if size < 32 or size & (size - 1):
            raise ValueError("img size must be a power of two >= 32 (the reference raises below 32: SURVEY §0)")
        self.B, self.S, self.c = batch, size, n_critic
        self.dev = torch.device(device)
        self.code = _lib.dtype_code(dtype)                          # storage dtype of activations / every non-conv kernel
        self.mma = _lib.mma_code(dtype)                             # what the conv entry points get: != code in the split modes
        self.mode = _lib.dtype_name(self.mma)
        self.T = _lib.torch_dtype(self.code)
        self.lr, self.betas = lr, betas
        self.delta_scale, self.lambda_gp, self.lambda_iou = delta_scale, lambda_gp, lambda_iou
        self.seed = seed
        self.allreduce = allreduce
        # Static loss scales of the fp16 mode (1 elsewhere: bf16 and fp32 have fp32's exponent range).  The backward seeds are
        # 1/(B hw) for the critic (hw score positions per sample) and ~1/(B S^2) per element for the generator (EIoU mean over
        # the batch, then the average pool): unscaled the 16-bit gradient tensors sit at 1e-4 / 1e-7 per element, at and
        # far below fp16's smallest normal 6e-5.  The generator's scale B S^2 makes its seeds O(1) (measured peak 2e2 over 1500
        # iterations).  The critic's tensors have a heavy tail -- dzs is rstd (up to 316 on a 2x2 map) times the incoming
        # gradient: un-scaled peaks of 10..125 in 18 000 iterations (tools/nan_hunt.py, bf16) against a median of 1e-4 -- so its
        # scale is B hw / 8 (32 at B=256, 32x32: peak ~4e3 of the 65504 ceiling, a few % of the entries below the normal
        # range), and fp16 stores saturate instead of producing inf (common.h).  Powers of two: exact.
        f16 = self.code == _lib.F16
        # fp16 halves of fp32 operands (fp16x3): the same range argument and the same scales as the fp16 mode.  Nothing is stored in
        # 16 bits here, and the split does not clamp (an operand beyond 65504 becomes inf, its residual -inf, the product NaN: loud),
        # so the scale must keep the heavy tail of the critic's gradient tensors under the ceiling with margin: with B hw (256 at the
        # bench shape: 8x this) one nan_hunt trial in four went non-finite at iteration 1240 (peak |dzs| 8208 seen in a finite one),
        # with B hw / 8 the peaks of four 1500-iteration trials stay under 1.1e3 and the measured errors are unchanged.
        x3h = self.mma == _lib.F32_F16X3
        pow2 = lambda v: float(2 ** round(math.log2(max(v, 1.0))))
        hw5 = (size // 16 - 1) ** 2
        self.loss_scale_d = float(os.environ.get("GCSSL_LOSS_SCALE_D", pow2(batch * hw5 / 8.0) if (f16 or x3h) else 1.0))
        self.loss_scale_g = float(os.environ.get("GCSSL_LOSS_SCALE_G", pow2(batch * size * size) if (f16 or x3h) else 1.0))
        self.keep_clipped_grads = keep_clipped_grads     # write g*clip_coef back like clip_grad_norm_ does (not needed to step)
        _lib.lib()                                                  # fail loudly now if the HIP library is missing
        _lib.call_nostream("gcssl_init")                            # dynamic-LDS opt-ins, before any graph capture
        dev = self.dev
        f32 = dict(device=dev, dtype=torch.float32)
        # spectral_norm=False: Discriminator(spectral_norm=False) -- plain convs under the keys model.N.weight, sigma = 1
        self.spectral_norm = bool(spectral_norm)
        self.d_wkey = (lambda i: f"model.{i}.weight_orig") if self.spectral_norm else (lambda i: f"model.{i}.weight")
        self.d_keys = D_PARAM_KEYS if self.spectral_norm else [k.replace("weight_orig", "weight") for k in D_PARAM_KEYS]
        self.D = FlatParams({k: sd_d[k].to(dev, torch.float32) for k in self.d_keys}, self.d_keys, dev)
        g_keys = GS_PARAM_KEYS if generator_type == "simple" else G_PARAM_KEYS
        self.G = FlatParams({k: sd_g[k].to(dev, torch.float32) for k in g_keys}, g_keys, dev)
        if self.spectral_norm:
            self.u = [sd_d[f"model.{i}.weight_u"].to(dev, torch.float32).clone() for i in D_IDX]
            self.v = [sd_d[f"model.{i}.weight_v"].to(dev, torch.float32).clone() for i in D_IDX]
            self.sn = ops.SnState([self.D.views[self.d_wkey(i)] for i in D_IDX], self.u, self.v, 4, dev)   # slots 0-2: a critic step's three forwards; 3: the value-only forward
        else:
            self.u, self.v = [], []
            self.sn = _NoSpectralNorm(len(D_IDX), 4, dev)
        self._zcap, self._zkeep, self._splits, self._zfull = {}, [], {}, {}
        self.grad_slabs = os.environ.get("GCSSL_GRAD_SLABS", "1") != "0"
        # one batched generator forward per iteration (g_forward_all); GCSSL_BATCH_G=0: one forward per call, for A/B runs
        self.batch_g = os.environ.get("GCSSL_BATCH_G", "1") != "0"
        # the generator's up3 / up4 layers as ONE pixel-stationary launch each (transposed conv + InstanceNorm + ReLU + pool
        # sums: csrc/convt_fused.hip) where the shapes allow; bit 0 = up4, bit 1 = up3 (GCSSL_FUSED_UP=0: the unfused pair)
        self.fused_up = int(os.environ.get("GCSSL_FUSED_UP", "3"))
        # conv + InstanceNorm + LeakyReLU as ONE launch for the layers whose samples fit a GEMM tile whole (D.c2-c4, G.down2-4
        # at 32x32; csrc/igemm.hip FIN forms): the fp32 pre-norm tensor is never stored, the backward kernels rebuild xhat from
        # the 16-bit activation.  16-bit modes only; GCSSL_FIN=0 (read by the library too) restores the conv -> norm pairs.
        self._fin_cache = {}
        self._alloc()
        self.gen = SimpleGenerator(self) if generator_type == "simple" else None
        self._d_dirty = True
        self._g_dirty = True
        self._prep_d_batch = self._prep_g_batch = None
        self.alpha_buf = torch.empty(batch, **f32)
        # fp16 stores of gradient tensors saturate at +-65504 (common.h); every kernel that stores one counts the values it
        # clipped here: [0] the critic's backward / gradient-penalty chains, [1] the generator's backward.  Never reset by the
        # engine; bench.py reports the totals and the tests assert they stay 0 (if they do not, the static loss scale is too
        # large for that configuration).
        self.sat = torch.zeros(2, device=dev, dtype=torch.int32)
        self.sat_d, self.sat_g = self.sat[0:1], self.sat[1:2]
        self._red_d = self._red_g = None
        self._rep_sum = None
        # Independent branches of the iteration run on a side HIP stream (hipGraph capture turns them into parallel
        # graph branches): spectral-norm iterations + weight re-pack beside the no-grad generator forward, and every
        # layer's wgrad + split-K reduce beside the dependent dgrad -> norm-backward chain.  None of these kernels
        # fills 256 CUs on its own.
        self.overlap = int(overlap)
        self.side = [_side_stream(dev, f"fine{i}") for i in range(2)] if overlap else []
        self._in_g_branch = False
        # ONE coarse branch (round 3): the generator step's own work -- EIoU, backward through the head, the up and the down
        # path, its split-K reduction: ~45 launches, 0.45 ms, none of which fills the chip -- does not depend on the critic
        # (the WGAN term gives G no gradient) once the iteration's batched generator forward has run, and nothing
        # in the critic steps reads what it writes.  It runs on a second stream beside the critic steps (one fork, one join
        # per iteration; captured as a parallel graph branch).  Single-GPU schedule only: with data parallelism the same work
        # already sits under the critic's all-reduces.  GCSSL_OVERLAP_G=0 restores the serial order.
        # (2: also the first critic step's spectral-norm chain + re-pack beside the batched generator forward -- measured
        #  neutral, 120.5k vs 120.4k.  The older fine-grained `overlap` levels stay out of the generator's branch: a fork out of
        #  a captured branch crashed the ROCm 7.2 graph capture; beside it, on the critic's side only, they cost 4.5 %.)
        self.overlap_g = int(os.environ.get("GCSSL_OVERLAP_G", "1")) if allreduce is None else 0
        self.side_g = _side_stream(dev, "gen") if self.overlap_g else None
        self.side_sn = _side_stream(dev, "sn") if self.overlap_g >= 2 else None
        self.probe = None          # {label: {"events": [(start, stop)...], "flops": f}} when profiling is enabled
        self._wgrad_batch_on = os.environ.get("GCSSL_WGRAD_BATCH", "1") != "0"
        self._sn_defer = self.spectral_norm and os.environ.get("GCSSL_SN_DEFER", "1") != "0"     # (A/B knob)
        # (A/B knob, default OFF: measured 135.2 / 135.4k with it against 135.8 / 136.3k without -- the rider builds its tap table from
        #  the RAW weight with strided scalar loads and makes the re-pack launch longer than the launch it saves)
        self._c5_defer = os.environ.get("GCSSL_C5_DEFER", "0") != "0"
        self._wgrad_d = self._wgrad_gu = None
        self.step_log = self.delta_log = None                      # enable_step_log(): per-critic-step scalars / the generator step's delta, kept on the device
        self._k_cur = 0
        self.probe_repeats = 0

    # ------------------------------------------------------------------------------------------ side-stream branches
    def _on_side(self, fn, which: int = 0, level: int = 2):
        """Run fn() on side stream `which`, ordered after everything enqueued so far on the current stream.
        level 1 = the coarse branches (spectral norm + re-pack, the value-only critic forward), 2 = per-layer wgrads."""
        if self.overlap < level or self.probe is not None or self._in_g_branch:   # probing wants serial, attributable timings;
            fn()                                                                    # no forks out of the generator's branch
            return
        # (inside the generator branch of run_iteration the fine-grained branches fork from / join into THAT branch and use
        #  their own side stream: a stream shared with the critic's branches would order the two coarse branches against
        #  each other)
        st = self.side[which]
        main = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(main)
        st.wait_event(ev)
        with torch.cuda.stream(st):
            fn()

    def _join_side(self):
        """Make the current stream wait for everything enqueued on the side stream(s) of the current coarse branch."""
        if not self.overlap or self.probe is not None or self._in_g_branch:
            return
        for st in self.side:
            ev = torch.cuda.Event()
            ev.record(st)
            torch.cuda.current_stream().wait_event(ev)

    # ------------------------------------------------------------------------------------------ device-side logging
    def enable_step_log(self) -> None:
        """Keep what the reference logs per step (cgan/cgan_train_enhanced.py:335-337,372-374) ON THE DEVICE, so that a loop of graph
        replays needs no host sync per iteration: after an iteration ``step_log[k] = [mean D(real), mean D(fake), mean D(interp),
        gradient penalty]`` of critic step k and ``delta_log`` = the generator step's predicted deltas (copied out before the
        pipelined forward of the next iteration overwrites them).  One tiny launch per critic step + one per generator step,
        part of whatever is captured after this call."""
        if self.step_log is not None:
            return
        f32 = dict(device=self.dev, dtype=torch.float32)
        self.step_log = torch.zeros(self.c, 4, **f32)
        self.delta_log = torch.zeros(self.B, 4, **f32)
        self._log_step = [ops.ReplicaSum([(self.means, self.step_log[k, 0:3], 3, False), (self.gp_sum, self.step_log[k, 3:4], 1, False)], 1, 0)
                          for k in range(self.c)]
        self._log_delta = ops.ReplicaSum([(self.g_delta, self.delta_log, 4 * self.B, False)], 1, 0)

    # ------------------------------------------------------------------------------------------ in-situ kernel timing
    def enable_probe(self, on: bool = True, repeats: int = 0):
        """Bracket every MFMA conv launch with HIP events on the launch stream (eager mode only).
        repeats > 0: each launch is then issued `repeats` MORE times back to back inside a second event pair -- what the launch
        costs inside a chain of launches (the graph replay's situation) rather than behind an idle gap and an event packet.
        (Launches that accumulate into sums then over-count them: the engine's state after such a pass is scrap.)"""
        self.probe = {} if on else None
        self.probe_repeats = int(repeats) if on else 0

    def _conv(self, label: str, flops: float, fn, *args, _bytes=None, **kw):
        """_bytes: (algorithmic, stored) bytes of the launch when its argument list does not show them (fused launches)"""
        if self.mma != self.code and fn in ops.CONV_FNS:
            kw["dt"] = self.mma                                     # split-precision modes: fp32 tensors, 3 x 16-bit MFMA contraction
        if self.probe is None:
            return fn(*args, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn(*args, **kw)
        e1.record()
        e2 = None
        if self.probe_repeats:
            e2 = torch.cuda.Event(enable_timing=True)
            for _ in range(self.probe_repeats):
                fn(*args, **kw)
            e2.record()
        rec = self.probe.get(label)
        if rec is None:
            algo, stored = _bytes or _algorithmic_bytes(label, args, 4 if self.code == _lib.F32 else 2)
            rec = self.probe[label] = {"events": [], "flops": flops, "bytes": algo, "stored": stored,
                                       "kernel": ops.last_kernel(),      # the template expression the dispatcher launched
                                       "grid": ops.last_grid()}          # ... and its workgroup count (tools/prof_labels.py)
        rec["events"].append((e0, e1, e2))

    def probe_summary(self):
        """-> {label: (n_launches, mean_ms, flops_per_launch, algorithmic_bytes_per_launch, stored_bytes_per_launch, kernel,
        mean_ms_chained)} (synchronises).  kernel: the kernel template expression of the label's launch (gcssl_last_kernel);
        mean_ms_chained: per launch inside the back-to-back repeats (enable_probe(repeats=...)), else None.
        self.probe_grids (set here): {label: workgroups of the label's launch}."""
        torch.cuda.synchronize()

        def robust(ts):
            # an eager launch whose host-side enqueue stalls (GC pause, first-use lazy init) shows up as tens of ms between
            # its two events: average the samples within 3x the median (all of them, when nothing stalled)
            ts = sorted(ts)
            med = ts[len(ts) // 2] if ts else 0.0
            keep = [t for t in ts if t <= 3.0 * med] or ts
            return sum(keep) / max(len(keep), 1)
        out = {}
        self.probe_grids = {k: rec.get("grid", 0) for k, rec in (self.probe or {}).items()}
        for k, rec in (self.probe or {}).items():
            ev = rec["events"]
            rep = [b.elapsed_time(c) / self.probe_repeats for _, b, c in ev if c is not None]
            out[k] = (len(ev), robust([a.elapsed_time(b) for a, b, _ in ev]), rec["flops"], rec["bytes"], rec["stored"], rec["kernel"],
                      robust(rep) if rep else None)
        return out

    # ------------------------------------------------------------------------------------------ split-K slabs
    def _zbuf(self, kind: str, ns, hi: int, cin: int, cout: int, n_full: int) -> torch.Tensor:
        """fp32 pre-norm buffer [n_full][H][W][C] of a conv -> InstanceNorm pair, with room behind it for the K-split
        partial-sum slabs of the producing conv at the batch sizes `ns` the engine launches it with (kind 'fwd': conv of a
        hi x hi input, output hi/2; 'dgrad': transposed conv to a hi x hi output of `cin` channels).  The view returned is
        slab 0; _split(...) says how many slabs a launch uses."""
        ho, c = (hi // 2, cout) if kind == "fwd" else (hi, cin)
        need = n_full
        for n in ns:
            need = max(need, n * ops.conv_splits(kind, self.mma, n, hi, cin, cout))
        buf = torch.empty(need, ho, ho, c, device=self.dev, dtype=torch.float32)
        z = buf[:n_full]
        self._zcap[z.data_ptr()] = (buf.numel(), n_full)
        self._zfull[z.data_ptr()] = buf                            # (the whole allocation: the four-group critic forward writes 4B rows of it)
        self._zkeep.append(buf)
        return z

    def _split(self, kind: str, z: torch.Tensor, n: int, hi: int, cin: int, cout: int, max_hw: int = 256, grad: bool = False):
        """(nslab, slab_stride) for a launch over the first n samples of z: slab mode when the conv splits K and the slabs
        fit behind z, else (1, 0) = the atomic form."""
        key = (kind, z.data_ptr(), n)
        if grad and not self.grad_slabs:                           # (A/B knob GCSSL_GRAD_SLABS=0: gradient chains keep atomics)
            return 1, 0
        if key not in self._splits:
            ks = ops.conv_splits(kind, self.mma, n, hi, cin, cout)
            per = n * z.shape[1] * z.shape[2] * z.shape[3]
            # (the fused InstanceNorm kernels that add the slabs handle maps up to 16x16 -- 8x8 for the double backward;
            # larger ones keep the atomic form)
            cap, n_reg = self._zcap.get(z.data_ptr(), (0, -1))      # (a batch-slice view of a registered buffer has no slab room)
            ok = (ks > 1 and n_reg == z.shape[0] and ks * per <= cap and z.is_contiguous()
                  and z.shape[1] * z.shape[2] <= max_hw)
            self._splits[key] = (ks, per) if ok else (1, 0)
        return self._splits[key]

    def _fin(self, n: int, hi: int, cin: int, cout: int) -> bool:
        """does the fused conv + InstanceNorm + LeakyReLU launch serve a forward of these shapes?"""
        key = (n, hi, cin, cout)
        if key not in self._fin_cache:
            self._fin_cache[key] = self.code != _lib.F32 and ops.conv_in_act_ok(self.code, n, hi, cin, cout)
        return self._fin_cache[key]

    def _fin_x3(self, n: int, hi: int, cin: int, cout: int) -> bool:
        """split-precision modes: does the one-launch conv + InstanceNorm + LeakyReLU form on fp32 tensors serve these shapes?"""
        key = ("x3", n, hi, cin, cout)
        if key not in self._fin_cache:
            self._fin_cache[key] = self.mma != self.code and ops.conv_in_act_x3_ok(self.mma, n, hi, cin, cout)
        return self._fin_cache[key]

    def _actb(self, n: int, hi: int, cin: int, cout: int, with_sums: bool) -> bool:
        """does the one-launch dgrad + activation-backward form serve a data gradient of these shapes?"""
        key = ("actb", n, hi, cin, cout, with_sums)
        if key not in self._fin_cache:
            self._fin_cache[key] = self.mma != _lib.F32 and ops.conv_dgrad_act_bwd_ok(self.mma, n, hi, cin, cout, with_sums)     # (16-bit and split modes)
        return self._fin_cache[key]

    def _fwd_actb(self, n: int, hi: int, cin: int, cout: int) -> bool:
        """... and the one-launch first-layer conv + activation backward + dot of the reverse gradient-penalty chain?"""
        key = ("fwd_actb", n, hi, cin, cout)
        if key not in self._fin_cache:
            self._fin_cache[key] = self.mma != _lib.F32 and ops.conv_fwd_act_bwd_ok(self.mma, n, hi, cin, cout)      # (16-bit and split modes)
        return self._fin_cache[key]

    # ------------------------------------------------------------------------------------------ buffers
    def _alloc(self):
        B, S, T, dev = self.B, self.S, self.T, self.dev
        N3 = 3 * B

        def act(n, s, c, dt=T):
            return torch.empty(n, s, s, c, device=dev, dtype=dt)
        f32 = dict(device=dev, dtype=torch.float32)
        # ---- packed weights
        self.d_wf, self.d_wt = [], []
        for cin, cout in D_CH:
            cp = _pad8(cin)
            self.d_wf.append(torch.empty(cout, 16, cp, device=dev, dtype=T))
            self.d_wt.append(torch.empty(cp, 16, cout, device=dev, dtype=T))
        self.d_w5p = torch.empty(16, 512, **f32)
        unet = self.generator_type == "unet"                      # (the simple generator owns its buffers: gen_simple.py)
        self.gd_wf, self.gd_wt, self.gu_wf, self.gu_wt = [], [], [], []
        for cin, cout in (G_DOWN if unet else ()):
            cp = _pad8(cin)
            self.gd_wf.append(torch.empty(cout, 16, cp, device=dev, dtype=T))
            self.gd_wt.append(torch.empty(cp, 16, cout, device=dev, dtype=T))
        for cint, coutt in (G_UP if unet else ()):        # as a conv: Cout = CinT, Cin = CoutT
            self.gu_wf.append(torch.empty(cint, 16, coutt, device=dev, dtype=T))
            self.gu_wt.append(torch.empty(coutt, 16, cint, device=dev, dtype=T))
        # ---- critic, 3B batch
        # The operands of the two weight-gradient contributions of a critic step -- the 3B-sample backward (x = input
        # activations, dy = dzs) and the reverse of the B-sample GP chain (x = adjoint activations gt_a, dy = gb_zs) --
        # sit back to back in 4B-sample buffers, so ONE wgrad per layer contracts over both: half the launches, half the
        # split-K slabs to reduce.
        N4 = 4 * B
        x4 = act(N4, S, 8)
        x4.zero_()
        self.x4 = x4                                               # (all four groups: the critic forward with the value-only group batched in)
        self.x0, self.gt_x = x4[:N3], x4[N3:]
        sizes = [S // 2, S // 4, S // 8, S // 16]
        self.d_a4 = [act(N4, s, c) for s, (_, c) in zip(sizes, D_CH)]
        self.d_a = [t[:N3] for t in self.d_a4]
        self.d_x4 = [x4] + self.d_a4[:3]                                      # wgrad x operand of layer l
        # pre-InstanceNorm tensors are fp32 in both modes (z - mean(z) over 4..64 elements cancels a bf16 mantissa)
        # (each holds the K-split slabs of its producing conv when that conv splits: _zbuf)
        self.d_z = [None] + [self._zbuf("fwd", (N3, B, N4), S >> l, D_CH[l][0], D_CH[l][1], N3) for l in (1, 2, 3)]   # (N4: room for the four-group forward)
        self._d_zsrc = list(self.d_z)          # what the norm backward kernels read per layer: d_z (fp32) or, after a fused forward, d_a
        self.d_mean4 = [None] + [torch.empty(N4, c, **f32) for _, c in D_CH[1:]]
        self.d_rstd4 = [None] + [torch.empty(N4, c, **f32) for _, c in D_CH[1:]]
        self.d_mean = [None] + [t[:N3] for t in self.d_mean4[1:]]
        self.d_rstd = [None] + [t[:N3] for t in self.d_rstd4[1:]]
        self.h5 = sizes[3] - 1
        self.d_out4 = torch.empty(N4, self.h5, self.h5, **f32)
        self.d_out = self.d_out4[:N3]
        # GP chain (B samples)
        # gradient tensors that feed a norm/activation backward kernel are fp32 (norm.hip header); those that feed an
        # MFMA (gb_zs, gt_a, dzs) are in the compute dtype
        F32 = torch.float32
        # (those written by a K-split conv and read by a fused norm-backward kernel carry slab room: _zbuf)
        self.gb_a = [act(B, s, c, F32) for s, (_, c) in zip(sizes, D_CH)]     # d out / d a_l
        for l in (1, 2):
            self.gb_a[l] = self._zbuf("dgrad", (B,), S >> (l + 1), D_CH[l + 1][0], D_CH[l + 1][1], B)
        self.d_dzs4 = [act(N4, s, c) for s, (_, c) in zip(sizes, D_CH)]       # [3B: dzs of the batched backward | B: gb_zs]
        self.gb_zs = [t[N3:] for t in self.d_dzs4]                            # (d out / d z_l) * isig
        self.gb_x0 = torch.empty(B, S, S, 8, **f32)
        self.gp_nrm = torch.empty(B, **f32)
        self.gp_coef = torch.empty(B, **f32)
        self.gt_z = [act(B, sizes[0], 64, F32)] + [self._zbuf("fwd", (B,), S >> l, D_CH[l][0], D_CH[l][1], B) for l in (1, 2, 3)]
        self.gt_a = [t[N3:] for t in self.d_a4]
        self.zt = [None] + [act(B, s, c, F32) for s, (_, c) in zip(sizes[1:], D_CH[1:])]
        # backward of the 3B forward
        self.d_da = [act(N3, s, c, F32) for s, (_, c) in zip(sizes, D_CH)]
        # the head conv's data gradient is a constant per sample group (its seeds) times the weights: the backward's (3B samples)
        # and the GP chain's (B samples, seed 1) come from ONE launch into a 4B-sample buffer
        self.d_da4_3 = act(N4, sizes[3], 512, F32)
        self.d_da[3], self.gb_a[3] = self.d_da4_3[:N3], self.d_da4_3[N3:]
        self.d_dzs = [t[:N3] for t in self.d_dzs4]
        # scalars: [0:12] cdot[layer][group] = <dW_sn_k, W_orig>/sigma_k^2 (spectral-norm quotient rule), [12] gp_sum, [13] eiou acc, [14:17] group means, [17] wgan-G mean
        # ... followed by NREP replicas of the critic backward's striped sums (bias gradients of the four layers + the 12
        # cdot entries): hundreds of workgroups adding to one 256-byte bias vector serialise (70 us on a 17-us pass), so
        # each adds to replica (workgroup % NREP) and one gcssl_sum_replicas launch folds them.  One fill zeroes it all.
        self.NREP, self.REP_STRIDE = 32, 1024
        # (the three group means of a critic step's forward are the LAST three floats of the block and the value-only forward's mean
        #  the float behind it: one head-conv launch over four groups adds to all four -- StepEngine.d_main(with_g=True) -- and the
        #  per-step fill, which ends in front of that float, leaves it alone)
        self._zero_all = torch.zeros(32 + self.NREP * self.REP_STRIDE + 4, **f32)
        self.zero_blk = self._zero_all[:-1]
        self.scal = self.zero_blk[:32]
        self.rep = self.zero_blk[32:32 + self.NREP * self.REP_STRIDE].view(self.NREP, self.REP_STRIDE)
        self.rep_bias_off = [0, 64, 192, 448]                      # c1..c4 bias (64, 128, 256, 512 floats); cdot at 960
        self.cdot = self.scal[0:12].view(4, 3)
        self.gp_sum = self.scal[12:13]
        self.eiou_acc = torch.zeros(1, **f32)      # (its own tensor: the generator step's first half may run between two critic steps, whose d_main clears zero_blk)
        self.means = self._zero_all[-4:-1]
        self.wgan_mean = self._zero_all[-1:]
        # wgrad slabs: per D layer [chain splits + forward splits]
        self.d_slab, self.d_ns = [], []
        for l, (cin, cout) in enumerate(D_CH):
            hi = S >> l
            cp = _pad8(cin)
            nc, nf = 0, ops.wgrad_splits(N4, hi, hi, cp, cout)
            self.d_ns.append((nc, nf))
            self.d_slab.append(torch.empty(nc + nf, cout, 16, cp, **f32))
        # ---- generator forward: ONE set of buffers for all n_critic + 1 generator calls of an iteration (g_forward_all runs
        # them as one batch); group `c` -- the generator step's call, whose activations the backward reads -- is also what
        # the stand-alone calls (d_step / g_step / generator_delta) use, under the historic attribute names
        NG = self.c + 1
        n = NG * B
        ga = GFwd()
        ga.n = n
        ga.traw = torch.empty(n, 4, **f32)
        ga.delta = torch.empty(n, 4, **f32)
        self.gfa = ga
        self.g_traw, self.g_delta = ga.traw[self.c * B:], ga.delta[self.c * B:]
        self._gall_valid = False
        self.ws = torch.empty(2 * N3 * 512, **f32)                 # scratch of the large-map InstanceNorm backward
        self.ws_g = torch.empty(2 * B * 512, **f32)                # ... the generator's own (its backward may run beside the critic's)
        self.g_gdelta = torch.empty(B, 4, **f32)
        self.g_cal = torch.empty(B, 4, **f32)
        if not unet:
            return
        ga.cat3 = act(n, S // 2, 128)          # [up3 out (64) | d1 (64)]
        ga.cat2 = act(n, S // 4, 256)          # [up2 out (128) | d2 (128)]
        ga.cat1 = act(n, S // 8, 512)          # [up1 out (256) | d3 (256)]
        ga.d4 = act(n, S // 16, 512)
        z32 = torch.float32
        ga.zd = [None] + [self._zbuf("fwd", (n,), S >> k, G_DOWN[k][0], G_DOWN[k][1], n) for k in (1, 2, 3)]
        ga.zu = [self._zbuf("dgrad", (n,), S >> (3 - k), coutt, cint, n) for k, (cint, coutt) in enumerate(G_UP)]
        ga.u4 = act(n, S, 64)
        ga.dmean = [None] + [torch.empty(n, c, **f32) for c in (128, 256, 512)]
        ga.drstd = [None] + [torch.empty(n, c, **f32) for c in (128, 256, 512)]
        ustats = [torch.empty(2, n, c, **f32) for c in (256, 128, 64, 64)]      # mean | rstd back to back: one fill
        ga.umean = [t[0] for t in ustats]
        ga.urstd = [t[1] for t in ustats]
        shapes = [(n, S // 16, S // 16, 512), (n, S // 8, S // 8, 256), (n, S // 4, S // 4, 128)]
        sizes = [math.prod(sh) for sh in shapes]
        ga.maskbuf = torch.empty(sum(sizes), device=dev, dtype=torch.uint8)          # one launch draws all three
        ga.masks = [ga.maskbuf[sum(sizes[:j]):sum(sizes[:j + 1])].view(sh) for j, sh in enumerate(shapes)]
        ga.pooled = torch.empty(n, 64, **f32)
        ga.poolsum = torch.zeros(n, 64, **f32)                     # sum over H*W of u4 (up4's IN apply pass adds, the head consumes and clears)
        ga.ucnt = torch.zeros(n, 64, **f32)                        # fused up4: how many of u4's values are positive, per (n, c)
        ga.x8 = act(n, S, 8)                                       # NHWC8 input of the batched forward (pred, replicated per call)
        self.gf = gf = ga.group(self.c, B)
        self.g_cat3, self.g_cat2, self.g_cat1, self.g_d4, self.g_u4 = gf.cat3, gf.cat2, gf.cat1, gf.d4, gf.u4
        self.g_zd, self.g_zu, self.g_dmean, self.g_drstd = gf.zd, gf.zu, gf.dmean, gf.drstd
        self.g_umean, self.g_urstd, self.g_masks = gf.umean, gf.urstd, gf.masks
        self.g_pooled, self.g_poolsum, self.g_ucnt = gf.pooled, gf.poolsum, gf.ucnt
        self._up4_presums = False                                  # set by a fused up4 forward: cnt / pooled describe g_zu[3]
        self.g_apre4 = act(B, S // 16, 512)                        # fused down4: its activation WITHOUT dropout, backward samples only
        self._g_zsrc = list(self.g_zd)                             # per down layer: g_zd (fp32 z) or the 16-bit activation of a fused forward
        self.g_dab = torch.empty(B, 64, **f32)
        # generator backward
        self.g_dzu = [act(B, S // 8, 256), act(B, S // 4, 128), act(B, S // 2, 64), act(B, S, 64)]
        self.g_dcat3 = act(B, S // 2, 128, z32)
        self.g_dcat2 = act(B, S // 4, 256, z32)
        self.g_dcat1 = act(B, S // 8, 512, z32)
        self.g_dd4 = self._zbuf("fwd", (B,), S // 8, 256, 512, B)
        self.g_dzd = [act(B, S // 2, 64), act(B, S // 4, 128), act(B, S // 8, 256), act(B, S // 16, 512)]
        self.g_dd = [None] + [self._zbuf("dgrad", (B,), S >> k, G_DOWN[k][0], G_DOWN[k][1], B) for k in (1, 2, 3)]   # grad wrt d1..d3 via the down path
        self.g_slab_d, self.g_ns_d, self.g_slab_u, self.g_ns_u = [], [], [], []
        for k, (cin, cout) in enumerate(G_DOWN):
            hi = S >> k
            cp = _pad8(cin)
            ns = ops.wgrad_splits(B, hi, hi, cp, cout)
            self.g_ns_d.append(ns)
            self.g_slab_d.append(torch.empty(ns, cout, 16, cp, **f32))
        for k, (cint, coutt) in enumerate(G_UP):       # conv geometry: x hi-res [.., CoutT], dy lo-res [.., CinT]
            hi = S >> (3 - k)
            ns = ops.wgrad_splits(B, hi, hi, coutt, cint)
            self.g_ns_u.append(ns)
            self.g_slab_u.append(torch.empty(ns, cint, 16, coutt, **f32))

    def _reduce_batches(self):
        """One launch reduces the split-K slabs of every layer of a backward pass (+ the spectral-norm rank-1 terms)."""
        if self._red_d is None:
            sn = self.spectral_norm           # (without it: no quotient rule, only the slabs and the striped bias sums)
            self._red_d = ops.ReduceBatch(
                [dict(slab=self.d_slab[l], nsplit=sum(self.d_ns[l]), dw=self.D.gviews[self.d_wkey(i)], cout=cout,
                      cin=_pad8(cin), cin_real=cin, coef=self.cdot[l] if sn else None, u=self.sn.u_hist[l] if sn else None,
                      v=self.sn.v_hist[l] if sn else None, coef_rep=self.rep[0, 960 + 3 * l:963 + 3 * l] if sn else None,
                      bias_rep=self.rep[0, self.rep_bias_off[l]:self.rep_bias_off[l] + cout], dbias=self.D.gviews[f"model.{i}.bias"])
                 for l, ((cin, cout), i) in enumerate(zip(D_CH, D_IDX))], nrank=3 if sn else 0, nrep=self.NREP, rep_stride=self.REP_STRIDE)
            if self.gen is not None:                               # the simple generator reduces its own 3x3 slabs
                return self._red_d, None
            gW = self.G.gviews
            up = [dict(slab=self.g_slab_u[k], nsplit=self.g_ns_u[k],
                       dw=gW[f"up{k + 1}.model.0.weight" if k < 3 else "up4.0.weight"], cout=cint, cin=coutt, cin_real=coutt)
                  for k, (cint, coutt) in enumerate(G_UP)]
            down = [dict(slab=self.g_slab_d[k], nsplit=self.g_ns_d[k], dw=gW[f"down{k + 1}.model.0.weight"], cout=cout,
                         cin=_pad8(cin), cin_real=cin) for k, (cin, cout) in enumerate(G_DOWN)]
            self._red_g = ops.ReduceBatch(up + down)
        return self._red_d, self._red_g

    # ------------------------------------------------------------------------------------------ weights
    def _prep_d(self):
        if not self._d_dirty:
            return
        if self._prep_d_batch is None:
            self._prep_d_batch = ops.PrepBatch(
                [(self.D.views[self.d_wkey(i)], self.d_wf[l], self.d_wt[l], cout, cin, _pad8(cin))
                 for l, ((cin, cout), i) in enumerate(zip(D_CH, D_IDX))], self.mma,
                c5=(self.D.views["model.11.weight"], self.d_w5p))         # the head's fp32 re-pack rides on the same launch
        self._prep_d_batch.run()
        self._d_dirty = False

    def _prep_g(self):
        if not self._g_dirty:
            return
        if self.gen is not None:
            self.gen.prep()
            self._g_dirty = False
            return
        if self._prep_g_batch is None:
            layers = [(self.G.views[f"down{k + 1}.model.0.weight"], self.gd_wf[k], self.gd_wt[k], cout, cin, _pad8(cin))
                      for k, (cin, cout) in enumerate(G_DOWN)]
            for k, (cint, coutt) in enumerate(G_UP):
                key = f"up{k + 1}.model.0.weight" if k < 3 else "up4.0.weight"
                layers.append((self.G.views[key], self.gu_wf[k], self.gu_wt[k], cint, coutt, coutt))
            self._prep_g_batch = ops.PrepBatch(layers, self.mma)
        self._prep_g_batch.run()
        self._g_dirty = False

    # ------------------------------------------------------------------------------------------ critic forward
    def _d_forward(self, n: int, gscale_of_layer, group_n: int, x: Optional[torch.Tensor] = None, means=None, groups=0):
        """conv stack over the first n rows of the 3B buffers (input: x, default the first n rows of x0).  means (zeroed by
        the caller): += the mean score of each of `groups` equal sample groups, from the head conv's own launch."""
        x = self.x0[:n] if x is None else x
        wide = n > 3 * self.B                                      # four groups (d_main(with_g=True)): the 4B-sample buffers, fused layers only
        d_a = self.d_a4 if wide else self.d_a
        d_mean, d_rstd, d_out = (self.d_mean4, self.d_rstd4, self.d_out4) if wide else (self.d_mean, self.d_rstd, self.d_out)
        for l, ((cin, cout), i) in enumerate(zip(D_CH, D_IDX)):
            bias = self.D.views[f"model.{i}.bias"]
            fl = conv_flops(n, self.S >> l, cin, cout)
            if l == 0:
                self._conv(f"D.c1.fwd[n={n}]", fl, ops.conv_fwd, x, self.d_wf[0], d_a[0][:n], 8, cout, bias=bias,
                           gscale=gscale_of_layer(0), group_n=group_n, act=LRELU)
            elif self._fin(n, self.S >> l, cin, cout):
                # conv + InstanceNorm + LeakyReLU in one launch; the backward reads the activation instead of z
                self._conv(f"D.c{l + 1}.fwd[n={n}]", fl, ops.conv_in_act_fwd, d_a[l - 1][:n], self.d_wf[l],
                           d_a[l][:n], d_mean[l][:n], d_rstd[l][:n], cin, cout, bias=bias,
                           gscale=gscale_of_layer(l), group_n=group_n)
                self._d_zsrc[l] = self.d_a[l]
            elif wide and self._fin_x3(n, self.S >> l, cin, cout):
                zfull = self._zfull[self.d_z[l].data_ptr()]         # (d_z[l] is its first 3B rows: what the backward reads)
                self._conv(f"D.c{l + 1}.fwd[n={n}]", fl, ops.conv_in_act_x3_fwd, d_a[l - 1][:n], self.d_wf[l], zfull[:n],
                           d_a[l][:n], d_mean[l][:n], d_rstd[l][:n], cin, cout, bias=bias, gscale=gscale_of_layer(l), group_n=group_n)
                self._d_zsrc[l] = self.d_z[l]
            elif wide:
                # an unfused layer of the four-group forward (the split modes' fourth layer at the bench shape): the conv adds its K
                # splits into z atomically (no slab room is registered for 4B rows), then the norm pass -- on the whole allocation
                zfull = self._zfull[self.d_z[l].data_ptr()]
                self._conv(f"D.c{l + 1}.fwd[n={n}]", fl, ops.conv_fwd, d_a[l - 1][:n], self.d_wf[l], zfull[:n], cin, cout, bias=bias,
                           gscale=gscale_of_layer(l), group_n=group_n, split_stride=0)
                ops.in_act_fwd(zfull[:n], d_a[l][:n], d_mean[l][:n], d_rstd[l][:n], cout, LRELU)
                self._d_zsrc[l] = self.d_z[l]
            elif self._fin_x3(n, self.S >> l, cin, cout):
                # split-precision modes: the same fusion on fp32 tensors; z is still stored (the fp32 backward kernels read it)
                self._conv(f"D.c{l + 1}.fwd[n={n}]", fl, ops.conv_in_act_x3_fwd, self.d_a[l - 1][:n], self.d_wf[l], self.d_z[l][:n],
                           self.d_a[l][:n], self.d_mean[l][:n], self.d_rstd[l][:n], cin, cout, bias=bias,
                           gscale=gscale_of_layer(l), group_n=group_n)
                self._d_zsrc[l] = self.d_z[l]
            else:
                ns, st = self._split("fwd", self.d_z[l], n, self.S >> l, cin, cout)
                self._conv(f"D.c{l + 1}.fwd[n={n}]", fl, ops.conv_fwd, self.d_a[l - 1][:n], self.d_wf[l],
                           self.d_z[l][:n], cin, cout, bias=bias, gscale=gscale_of_layer(l), group_n=group_n, split_stride=st)
                ops.in_act_fwd(self.d_z[l][:n], self.d_a[l][:n], self.d_mean[l][:n], self.d_rstd[l][:n], cout, LRELU,
                               nslab=ns, slab_stride=st)
                self._d_zsrc[l] = self.d_z[l]
        ops.c5_fwd(d_a[3][:n], self.d_w5p, d_out[:n], group_mean=means, groups=groups)

    def gbatch_ok(self) -> bool:
        """can the generator step's value-only critic forward be a fourth group of the next critic forward (d_main(with_g=True))?
        Every normalised layer takes the fused conv + InstanceNorm launch at 4B samples or has an fp32 pre-norm buffer of 4B rows."""
        n4 = 4 * self.B
        return self.spectral_norm and all(
            self._fin(n4, self.S >> l, *D_CH[l]) or self._zfull[self.d_z[l].data_ptr()].shape[0] >= n4 for l in (1, 2, 3))

    def critic_scores(self, pred: torch.Tensor, other: torch.Tensor, train: bool = True) -> torch.Tensor:
        """Discriminator.forward(pred, other) (cgan/models.py:255-258) for one (B,3,S,S) pair -> (B,1,h,w).
        train=True advances the spectral-norm u,v by one power iteration like the reference's train-mode forward."""
        B = pred.shape[0]
        assert B <= 3 * self.B
        self.sn.iterate(0, iterate=train)
        self._prep_d()
        ops.pack_pair(pred, other, self.x0[:B])
        self._d_forward(B, lambda l: self.sn.isig[l, 0:1], B)
        return self.d_out[:B].reshape(B, 1, self.h5, self.h5)

    # ------------------------------------------------------------------------------------------ generator forward
    def _set_masks(self, masks: Optional[Sequence[torch.Tensor]], phase: int = 0):
        """masks given as NCHW keep-masks (fixture/parity mode) or None -> drawn on the device.  The draw is keyed by
        (seed, phase, generator step count): the count lives on the device (Adam state), so graph replays draw fresh
        masks, and it is constant within an iteration, so the draw does not depend on where d_pre runs relative to the
        critic updates; phase 10+k = critic step k, 1 = the generator step, 2 = generator_delta()."""
        if self.gen is not None:
            self.gen.set_masks(masks, phase)
        elif masks is None:
            for j, m in enumerate(self.g_masks):                   # (views into the all-calls buffers: one launch each)
                ops.dropout_mask_gen(m, self.seed * 131 + phase + 7919 * j, self.G.state)
        else:
            for m, src in zip(self.g_masks, masks):
                m.copy_(src.permute(0, 2, 3, 1))

    def _g_forward(self, x8: torch.Tensor, train: bool = True):
        """GeneratorUNet.forward (cgan/models.py:125-141) on an NHWC8 input whose first 3 channels are pred."""
        if self.gen is not None:
            return self.gen.forward(x8, train)
        return self._unet_forward(x8, train, self.gf)

    def _unet_forward(self, x8: torch.Tensor, train: bool, f: GFwd) -> torch.Tensor:
        """The U-Net forward on the buffer set `f`: the generator step's group (B samples), or all n_critic + 1 calls of an
        iteration at once (g_forward_all)."""
        n, S = f.n, self.S
        tag = "" if n == self.B else f"[n={n}]"
        d1, d2, d3 = f.cat3[..., 64:], f.cat2[..., 128:], f.cat1[..., 256:]
        mk = f.masks if train else [None, None, None]
        self._conv(f"G.down1.fwd{tag}", conv_flops(n, S, 3, 64), ops.conv_fwd, x8, self.gd_wf[0], d1, 8, 64, act=LRELU)
        dins, douts, dmask = [None, d1, d2, d3], [None, d2, d3, f.d4], [None, None, None, mk[0]]
        bwd_here = f.n == self.B or f is self.gfa                 # does this pass hold the generator step's group (its last B samples)?
        for k in (1, 2, 3):
            cin, cout = G_DOWN[k]
            if self._fin(n, S >> k, cin, cout):
                # conv + InstanceNorm + LeakyReLU (+ dropout) in one launch.  down4's output is masked, so the samples that have
                # a backward pass (the last B of this batch) also store the un-masked activation for it.
                apre = self.g_apre4 if (k == 3 and dmask[k] is not None and bwd_here) else None
                self._conv(f"G.down{k + 1}.fwd{tag}", conv_flops(n, S >> k, cin, cout), ops.conv_in_act_fwd, dins[k],
                           self.gd_wf[k], douts[k], f.dmean[k], f.drstd[k], cin, cout, mask=dmask[k], apre=apre,
                           apre_n0=n - self.B)
                if bwd_here:
                    grp = douts[k][n - self.B:]
                    self._g_zsrc[k] = self.g_apre4 if apre is not None else grp
                continue
            if bwd_here:
                self._g_zsrc[k] = self.g_zd[k]
            if dmask[k] is None and self._fin_x3(n, S >> k, cin, cout):          # (split-precision modes; down4's output is masked: unfused)
                self._conv(f"G.down{k + 1}.fwd{tag}", conv_flops(n, S >> k, cin, cout), ops.conv_in_act_x3_fwd, dins[k], self.gd_wf[k],
                           f.zd[k], douts[k], f.dmean[k], f.drstd[k], cin, cout)
                continue
            ns, st = self._split("fwd", f.zd[k], n, S >> k, cin, cout)
            self._conv(f"G.down{k + 1}.fwd{tag}", conv_flops(n, S >> k, cin, cout), ops.conv_fwd, dins[k], self.gd_wf[k],
                       f.zd[k], cin, cout, split_stride=st)
            ops.in_act_fwd(f.zd[k], douts[k], f.dmean[k], f.drstd[k], cout, LRELU, mask=dmask[k],
                           nslab=ns, slab_stride=st)
        ins = [f.d4, f.cat1, f.cat2, f.cat3]
        outs = [f.cat1[..., :256], f.cat2[..., :128], f.cat3[..., :64], f.u4]
        for k, (cint, coutt) in enumerate(G_UP):
            hin = S >> (4 - k)                                        # input map of the transposed conv
            if k >= 2 and (self.fused_up >> (3 - k)) & 1 and ops.convt_fused_ok(self.code, n, hin, coutt):
                # one launch: conv + statistics + ReLU (+ pool sums for up4, whose activation nobody reads: only its sums
                # feed the head).  The fp32 pre-norm values are kept for the samples that have a backward pass: all of them in
                # a single-call forward, the generator step's group in the iteration's batched forward.
                es, hw_o = 2, 4 * hin * hin
                algo = (ins[k].numel() + self.gu_wt[k].numel() + n * hw_o * coutt) * es
                z_n0 = 0 if n == self.B else n - self.B
                stored = (ins[k].numel() + self.gu_wt[k].numel() + (n * hw_o * coutt if k < 3 else 0)) * es + (n - z_n0) * hw_o * coutt * 4
                self._conv(f"G.up{k + 1}.fwd{tag}", conv_flops(n, 2 * hin, coutt, cint), ops.convt_in_relu_fwd, ins[k],
                           self.gu_wt[k], f.umean[k], f.urstd[k], cint, z32=f.zu[k], z_n0=z_n0, a=outs[k] if k < 3 else None,
                           pool=f.poolsum if k == 3 else None, cnt=f.ucnt if k == 3 else None, _bytes=(algo, stored))
                if k == 3:
                    self._up4_presums = True
                continue
            ns, st = self._split("dgrad", f.zu[k], n, S >> (3 - k), coutt, cint) if k < 3 else (1, 0)
            self._conv(f"G.up{k + 1}.fwd{tag}", conv_flops(n, S >> (3 - k), coutt, cint), ops.conv_dgrad, ins[k],
                       self.gu_wt[k], f.zu[k], coutt, cint, split_stride=st)
            # (up4's activation feeds nothing but the head's average pool and its backward reads z: where the launch can do
            # without the store -- 32x32 maps -- it is not written)
            skip_a = k == 3 and ns == 1 and ops.in_act_fwd_pool_only_ok(S * S, coutt)
            ops.in_act_fwd(f.zu[k], None if skip_a else outs[k], f.umean[k], f.urstd[k], coutt, RELU,
                           mask=mk[k + 1] if k < 2 else None, pool=f.poolsum if k == 3 else None,
                           nslab=ns, slab_stride=st)
            if k == 3:
                self._up4_presums = False
        ops.pool_fc_tanh_fwd(f.u4, self.G.views["fc_delta.1.weight"], self.G.views["fc_delta.1.bias"],
                             self.delta_scale, f.pooled, f.traw, f.delta, pool_sum=f.poolsum)
        return f.delta

    def g_forward_all(self, pred: torch.Tensor, masks=None) -> None:
        """All n_critic + 1 generator forwards of an iteration as ONE batch (cgan/cgan_train_enhanced.py:311-312 per critic
        step and :348): they read the same input with the same weights -- G is only updated at :369 -- and differ in their
        Dropout draws, so the reference's separate calls are one (n_critic+1)*B-sample pass here, the way the three critic
        forwards of a step are one 3B-sample pass.  Same FLOPs; the small-M layers stop being latency-bound.  d_pre(k) then
        takes its delta from group k, g_main its activations from group n_critic.  masks: per call a triple (fixtures)
        or None (drawn on the device, one launch)."""
        B, fa = self.B, self.gfa
        self._prep_g()
        if self.gen is not None:
            self.gen.forward_all(pred, masks)
            self._gall_valid = True
            return
        ops.pack_pair(pred, None, fa.x8, reps=self.c + 1)          # the same input for every call
        if masks is None:
            ops.dropout_mask_gen(fa.maskbuf, self.seed * 131 + 20, self.G.state)
        else:
            for g, triple in enumerate(masks):
                for m, src in zip(fa.masks, triple):
                    m[g * B:(g + 1) * B].copy_(src.permute(0, 2, 3, 1))
        self._unet_forward(fa.x8, True, fa)
        self._gall_valid = True

    def generator_delta(self, pred: torch.Tensor, masks=None, train: bool = True) -> torch.Tensor:
        """GeneratorUNet.forward(pred) -> (B,4) delta."""
        self._prep_g()
        self._gall_valid = False                      # (this forward overwrites group n_critic of a pending batched forward: a
        #                                                pipelined GraphedIteration re-runs its prologue on the next replay)
        ops.pack_pair(pred, None, self.gt_x)          # gt_x doubles as the NHWC8 staging buffer outside a D step
        if train:
            self._set_masks(masks, 2)
        return self._g_forward(self.gt_x, train).clone()

    # ------------------------------------------------------------------------------------------ D step
    def d_step(self, pred, gt, refine_fn, k: int, alpha: Optional[torch.Tensor], masks) -> None:
        """One critic update (cgan/cgan_train_enhanced.py:304-332)."""
        self.d_compute(pred, gt, refine_fn, k, alpha, masks)
        self.d_update()

    def allreduce_start(self, flat: torch.Tensor):
        """Begin averaging a flat gradient over the ranks without blocking this stream (dist.GradAverager.start);
        hooks that are plain callables run synchronously here."""
        start = getattr(self.allreduce, "start", None)
        if start is None:
            self.allreduce(flat)
            return None
        return start(flat)

    def allreduce_wait(self, handle, flat: torch.Tensor) -> None:
        """Order this stream behind a started exchange WITHOUT the update (GraphedIteration: the update itself is inside
        the next captured segment, with the factor _allreduce_finish would return baked in)."""
        self._allreduce_finish(handle, flat)

    def dp_grad_scale(self) -> float:
        world = getattr(self.allreduce, "world", None)
        return 1.0 / world if world else 1.0

    def _allreduce_finish(self, handle, flat: torch.Tensor) -> float:
        """Order this stream behind a started exchange; returns the factor the optimiser still has to apply to the (summed)
        gradient: dist.GradAverager leaves the 1/world to the fused clip+Adam launch, other hooks average themselves."""
        if handle is None:
            return 1.0
        world = getattr(self.allreduce, "world", None)
        if world is None:
            self.allreduce.finish(handle, flat)
            return 1.0
        self.allreduce.finish(handle, flat, scale=False)
        return 1.0 / world

    def d_update(self, handle="sync") -> None:
        """all-reduce (data parallel) -> clip_grad_norm_(1.0) -> Adam  (:331-332).  handle: what allreduce_start
        returned when the exchange was started earlier; "sync" = do it here."""
        self._join_side()
        gs = 1.0
        if isinstance(handle, float):                             # exchange already waited for: only the averaging factor is left
            gs = handle
        elif handle != "sync":
            gs = self._allreduce_finish(handle, self.D.g)
        elif self.allreduce is not None:
            self.allreduce(self.D.g)
        ops.clip_adam(self.D.p, self.D.g, self.D.m, self.D.v, self.D.state, self.lr, self.betas[0], self.betas[1],
                      write_clipped=1 if self.keep_clipped_grads else 2, grad_scale=gs / self.loss_scale_d)   # 2: the update also re-zeroes the bucket
        self.D.grads_zero = not self.keep_clipped_grads
        self._d_dirty = True

    def d_compute(self, pred, gt, refine_fn, k: int, alpha: Optional[torch.Tensor], masks) -> None:
        """Forward passes, gradient penalty and all critic gradients of one critic step (:305-330); pure kernel
        launches on the current stream (hipGraph capturable).  Two parts: d_pre does not depend on the critic's weights
        (with data parallelism it runs under the previous critic step's all-reduce), d_main does."""
        if self.overlap >= 1 and self.probe is None:              # SN + re-pack on the side stream, under the generator forward
            self._on_side(self._sn_and_prep, 0, 1)
            self.d_pre(pred, gt, refine_fn, k, alpha, masks)
            self.d_main(sn_done=True)
        else:
            self.d_pre(pred, gt, refine_fn, k, alpha, masks)
            self.d_main()

    def d_pre(self, pred, gt, refine_fn, k: int, alpha: Optional[torch.Tensor], masks) -> None:
        """The generator side of a critic step: no-grad train-mode forward (:311-312), the re-crop (:313-315), alpha
        (cgan/losses.py:199) and the packed fake / interpolated groups."""
        B = self.B
        I = slice(2 * B, 3 * B)
        self._k_cur = k                                            # (host-side: which critic step the next d_main belongs to)
        real_out = None
        if self._gall_valid:                                       # this call was part of the iteration's batched forward
            if k == 0:                                             # (pred and gt are the iteration's: the real group is packed once,
                real_out = self.x0[:B]                             #  in the launch that packs the fake and interpolated groups)
            delta_det = self.gfa.delta[k * B:(k + 1) * B]
        else:
            ops.pack_pair(pred, gt, self.x0[:B])                   # real group; channels 0-2 = pred are G's input too
            self._prep_g()
            self._set_masks(masks, 10 + k)                         # keyed by (seed, step index, critic step count)
            delta_det = self._g_forward(self.x0[:B], True)
        refined = refine_fn(delta_det, k)                          # :313-315
        # fake (pred, refined) and interpolated groups in one pass; alpha given (parity runs) or drawn in the kernel
        ops.pack_fake_interp(pred, gt, refined, alpha, self.x0[B:2 * B], self.x0[I], seed=self.seed * 131 + 7 + 16 * k,
                             counter=self.G.state, out_real=real_out)

    def _sn_and_prep(self, zero_wgan: bool = False) -> None:
        # real, fake, interp forwards each iterate once: three chained power iterations (slots 0..2), whose closing launch also
        # clears the per-step scalars + striped-sum replicas (zero_blk) -- no fill launch of its own
        # (the chain's closing step rides on the re-pack launch that follows it when there is one: ops.SpectralNorm.iterate)
        # zero_wgan: the fill also covers wgan_mean (the float behind the block): a four-group forward follows
        self.sn.iterate(0, 3, zero=self._zero_all if zero_wgan else self.zero_blk, defer_finish=self._sn_defer and self._d_dirty)
        self._prep_d()

    def d_main(self, sn_done: bool = False, with_g: bool = False, zero_wgan: bool = False) -> None:
        """The critic side: spectral-norm iterations, the 3B-sample forward, gradient penalty, all gradients.
        with_g (GraphedIteration, batch_g_critic): the PREVIOUS iteration's value-only forward (g_critic: train-mode D on (pred,
        refined_G), cgan/cgan_train_enhanced.py:361-362) rides on this step's forward as a fourth group -- x4[3B:] holds its packed
        input.  The critic's weights have not changed since that iteration's last update, so its power iteration (slot 3) and this
        step's three (slots 0-2) are consecutive iterations on the same W, exactly the reference's sequence; its mean lands in
        wgan_mean, next to the three group means."""
        B, S, N3 = self.B, self.S, 3 * self.B
        assert not (with_g and sn_done)
        if with_g:
            self.sn.iterate(3, True)                              # the value-only forward's power iteration comes first
        I = slice(2 * B, 3 * B)
        isig = self.sn.isig
        hw = self.h5 * self.h5
        ls = self.loss_scale_d
        seeds = (-ls / (B * hw), ls / (B * hw), 0.0)
        # the head conv's data gradient for the constant seeds depends on nothing but the head's weight: where this call re-packs the
        # weights it rides on that launch (gcssl_conv4x4s1_c1_dgrad_defer), else it is launched behind the forward as before
        c5_rides = (not sn_done) and self._d_dirty and self._c5_defer and self.d_da4_3.dtype == torch.float32
        if c5_rides:
            ops.c5_dgrad_defer(self.d_da4_3, self.D.views["model.11.weight"], consts=(seeds[0], seeds[1], seeds[2], 1.0), group_n=B)
        if not sn_done:
            self._sn_and_prep(zero_wgan=with_g or zero_wgan)      # (zero_wgan: a replay with nothing owed leaves wgan_mean at 0, not stale)
        if not self.D.grads_zero:                                 # optimizer.zero_grad() (:305) unless the last update did it
            self.D.g.zero_()
        self.D.grads_zero = False
        self._join_side()                                         # sigma, u/v history, packed weights ready; zero_blk cleared
        if with_g:
            self._d_forward(4 * B, lambda l: isig[l], B, x=self.x4, means=self._zero_all[-4:], groups=4)
        else:
            self._d_forward(N3, lambda l: isig[l, :3], B, means=self.means, groups=3)      # (+ the three batch means, :327)
        # ---- GP first-order chain on the interpolated group (cgan/losses.py:213-220); its seed (1 per score) and the seeds of
        # the batched backward of the three forwards, -1/(B hw), +1/(B hw), 0 (:327-330), leave the head conv in one launch
        if not c5_rides:
            ops.c5_dgrad(self.d_da4_3, self.d_w5p, consts=(seeds[0], seeds[1], seeds[2], 1.0), group_n=B)
        ns, st = 1, 0                                             # K-split slabs of the conv that produced gb_a[l]
        gp_c1_done = False
        for l in (3, 2, 1):
            cin, cout = D_CH[l]
            ops.in_act_bwd(self._d_zsrc[l][I], self.d_mean[l][I], self.d_rstd[l][I], self.gb_zs[l], cout, LRELU,
                           da=self.gb_a[l], gscale=isig[l, 2:3], group_n=B, ws=self.ws, da_nslab=ns, da_slab_stride=st, sat=self.sat_d)
            if l == 1 and self._actb(B, S >> 1, cin, cout, False):
                # c2's data gradient with c1's LeakyReLU backward in its epilogue: gb_zs[0] directly, no fp32 gb_a[0]
                self._conv("D.c2.gp_dgrad", conv_flops(B, S >> 1, cin, cout), ops.conv_dgrad_act_bwd, self.gb_zs[1], self.d_wt[1],
                           self.d_a[0][I], self.gb_zs[0], cin, cout, gscale=isig[0, 2:3], group_n=B, sat=self.sat_d)
                gp_c1_done = True
                continue
            ns, st = self._split("dgrad", self.gb_a[l - 1], B, S >> l, cin, cout, grad=True) if l > 1 else (1, 0)
            self._conv(f"D.c{l + 1}.gp_dgrad", conv_flops(B, S >> l, cin, cout), ops.conv_dgrad, self.gb_zs[l],
                       self.d_wt[l], self.gb_a[l - 1], cin, cout, split_stride=st)
        if not gp_c1_done:
            ops.act_bwd(self.gb_a[0], self.d_a[0][I], self.gb_zs[0], 64, gscale=isig[0, 2:3], group_n=B, sat=self.sat_d)
        self._conv("D.c1.gp_dgrad", conv_flops(B, S, 6, 64), ops.conv_dgrad, self.gb_zs[0], self.d_wt[0], self.gb_x0, 8, 64)
        # :223-231, and the seed of the reverse pass (gb_x0 * coef, the create_graph=True part of d_loss.backward(), :330)
        # (lambda_gp only enters the adjoint seed coef/scaled, not gp_sum: the critic's loss scale rides on it)
        ops.gp_norm(self.gb_x0, B, self.lambda_gp * self.loss_scale_d, self.gp_nrm, self.gp_coef, self.gp_sum, scaled=self.gt_x, sat=self.sat_d)
        src = self.gt_x
        for l, (cin, cout) in enumerate(D_CH):
            cp = _pad8(cin)
            fl = conv_flops(B, S >> l, cin, cout)
            if l == 0 and self._fwd_actb(B, S, cp, cout):
                # conv + c1's LeakyReLU backward + the <gb_zs, gt_z> spectral-norm term as one launch: no fp32 gt_z[0]
                self._conv("D.c1.gp_rev_fwd", fl, ops.conv_fwd_act_bwd, src, self.d_wf[0], self.d_a[0][I], self.gt_a[0], cp, cout,
                           gscale=isig[0, 2:3], group_n=B, dotx=self.gb_zs[0], dot_out=self.cdot[0, 2:3], sat=self.sat_d)
                src = self.gt_a[0]
                continue
            ns, st = self._split("fwd", self.gt_z[l], B, S >> l, cin, cout, max_hw=64, grad=True) if l > 0 else (1, 0)
            self._conv(f"D.c{l + 1}.gp_rev_fwd", fl, ops.conv_fwd, src, self.d_wf[l], self.gt_z[l], cp, cout,
                       gscale=isig[l, 2:3], group_n=B, split_stride=st)
            # (the weight gradient of this chain, src x gb_zs[l], is contracted together with the batched backward's below)
            if l == 0:          # (+ the <gb_zs, gt_z> spectral-norm term of this norm-less layer, in the same pass)
                ops.act_bwd(self.gt_z[0], self.d_a[0][I], self.gt_a[0], 64, sat=self.sat_d, dotx=self.gb_zs[0], dot_out=self.cdot[0, 2:3])
            else:
                ops.in_dbl_bwd(self.gb_a[l], self.gt_z[l], self.gb_zs[l], self._d_zsrc[l][I], self.d_mean[l][I],
                               self.d_rstd[l][I], self.gt_a[l], self.zt[l], cout, LRELU, cdot=self.cdot[l, 2:3],
                               q_nslab=ns, q_slab_stride=st, sat=self.sat_d)
            src = self.gt_a[l]
        gw5 = self.D.gviews["model.11.weight"].view(512, 16)
        # ---- backward of the three forwards, batched: seeds -1/(B hw), +1/(B hw), 0  (:327-330)
        # the head's weight gradient of both parts in one launch over the 4B-sample buffer: seeds of the three forwards, and 1 for
        # the reverse GP chain's adjoint activations gt_a (the derivative of the first-order seed w5 * 1)
        ops.c5_wgrad(self.d_a4[3], gw5, 512, consts=(seeds[0], seeds[1], seeds[2], 1.0), group_n=B)
        c1_done = False
        # the weight gradients of c2-c4 as ONE launch behind the dgrad chain (they depend on nothing but their layer's dzs / x;
        # GCSSL_WGRAD_BATCH=0: one launch per layer, beside the chain as before)
        wgrad_batch = self._wgrad_batch_on
        wgrad_last, deferred = os.environ.get("GCSSL_WGRAD_LAST", "0") != "0", []
        for l in (3, 2, 1, 0):
            cin, cout = D_CH[l]
            cp = _pad8(cin)
            i = D_IDX[l]
            bias, gbias = self.D.views[f"model.{i}.bias"], self.D.gviews[f"model.{i}.bias"]
            rb = self.rep[0, self.rep_bias_off[l]:self.rep_bias_off[l] + cout]          # replica 0 of this layer's bias sums
            rc = self.rep[0, 960 + 3 * l:963 + 3 * l]                                    # ... and of its three cdot entries
            if l > 0:
                ops.in_act_bwd(self._d_zsrc[l][:N3], self.d_mean[l], self.d_rstd[l], self.d_dzs[l], cout, LRELU,
                               da=self.d_da[l], zt=self.zt[l], zt_n0=2 * B, gscale=isig[l], group_n=B, bias=bias,
                               dbias=rb, cdot=rc, ws=self.ws, nrep=self.NREP, rep_stride=self.REP_STRIDE, sat=self.sat_d)
            elif not c1_done:
                ops.act_bwd(self.d_da[0], self.d_a[0], self.d_dzs[0], 64, gscale=isig[0], group_n=B, bias=bias,
                            dbias=rb, cdot=rc, nrep=self.NREP, rep_stride=self.REP_STRIDE, sat=self.sat_d)
            fl = conv_flops(N3, S >> l, cin, cout)

            def wgrad_branch(l=l, cout=cout, cp=cp, fl4=conv_flops(4 * B, S >> l, cin, cout)):
                self._conv(f"D.c{l + 1}.wgrad", fl4, ops.conv_wgrad, self.d_x4[l], self.d_dzs4[l], self.d_slab[l], cp, cout)
            if wgrad_batch and l >= 1:
                pass                                              # (c2-c4: the batched launch behind the chain)
            elif wgrad_last:
                deferred.append(wgrad_branch)                     # (A/B: all weight gradients behind the dgrad chain)
            else:
                self._on_side(wgrad_branch)                       # beside the dgrad -> norm-backward chain
            if l == 1 and self._actb(N3, S >> 1, cin, cout, True):
                # c2's data gradient with c1's LeakyReLU backward (+ its bias-gradient and spectral-norm sums) in the epilogue:
                # d_dzs[0] directly; the 3B x 16x16x64 fp32 d_da[0] is neither written nor read back
                i0 = D_IDX[0]
                self._conv("D.c2.dgrad", fl, ops.conv_dgrad_act_bwd, self.d_dzs[1], self.d_wt[1], self.d_a[0], self.d_dzs[0], cin, cout,
                           gscale=isig[0], group_n=B, bias=self.D.views[f"model.{i0}.bias"],
                           dbias=self.rep[0, self.rep_bias_off[0]:self.rep_bias_off[0] + 64], cdot=self.rep[0, 960:963],
                           nrep=self.NREP, rep_stride=self.REP_STRIDE, sat=self.sat_d)
                c1_done = True
            elif l > 0:
                self._conv(f"D.c{l + 1}.dgrad", fl, ops.conv_dgrad, self.d_dzs[l], self.d_wt[l], self.d_da[l - 1], cin, cout)
        for fn in deferred:
            fn()
        if wgrad_batch:
            if self._wgrad_d is None:
                lay = [(self.d_x4[l], self.d_dzs4[l], self.d_slab[l], _pad8(D_CH[l][0]), D_CH[l][1]) for l in (3, 2, 1)]
                self._wgrad_d = ops.WgradBatch(lay)
                es = 4 if self.code == _lib.F32 else 2             # (bytes as engine._algorithmic_bytes counts a weight gradient)
                self._wgrad_d_bytes = (sum((x.numel() + dy.numel() + sl.numel() // sl.shape[0]) * es for x, dy, sl, _, _ in lay),
                                       sum(x.numel() * x.element_size() + dy.numel() * dy.element_size() + sl.numel() * 4 for x, dy, sl, _, _ in lay))
            self._conv("D.c2-4.wgrad", sum(conv_flops(4 * B, S >> l, *D_CH[l]) for l in (1, 2, 3)), ops.conv_wgrad_batch, self._wgrad_d,
                       _bytes=self._wgrad_d_bytes)
        self._join_side()                                         # all gradient branches are in before the segment ends
        # dW_orig = sum_k G_k / sigma_k - sum_k c_k u_k v_k^T for the four spectrally-normalised layers, one launch -- which also
        # folds the striped sums (bias gradients -> flat gradient; c_k = its GP-chain part in `cdot` + the replicas)
        self._reduce_batches()[0].run()
        if self.step_log is not None:
            self._log_step[self._k_cur].run()                     # this step's group means + gradient penalty -> step_log[k]

    # ------------------------------------------------------------------------------------------ G step
    def g_step(self, pred, delta_true, pred_box, refine_fn, masks) -> None:
        """The generator update (cgan/cgan_train_enhanced.py:345-369)."""
        self.g_compute(pred, delta_true, pred_box, refine_fn, masks)
        self.g_update()

    def g_update(self, handle="sync") -> None:
        self._join_side()
        gs = 1.0
        if isinstance(handle, float):
            gs = handle
        elif handle != "sync":
            gs = self._allreduce_finish(handle, self.G.g)
        elif self.allreduce is not None:
            self.allreduce(self.G.g)
        ops.clip_adam(self.G.p, self.G.g, self.G.m, self.G.v, self.G.state, self.lr, self.betas[0], self.betas[1],
                      write_clipped=1 if self.keep_clipped_grads else 2, grad_scale=gs / self.loss_scale_g)  # :368-369
        self.G.grads_zero = not self.keep_clipped_grads
        self._g_dirty = True
        self._gall_valid = False                                  # the iteration's batched generator forward is consumed

    def g_compute(self, pred, delta_true, pred_box, refine_fn, masks) -> None:
        """The generator step's launches (:345-366).  g_main (forward, EIoU, backward: the gradient) and g_critic (the
        value-only critic forward of :361-362) are independent of each other; with data parallelism g_critic runs under
        the generator gradient's all-reduce."""
        self.g_main(pred, delta_true, pred_box, refine_fn, masks)
        self.g_critic(pred)

    def g_critic(self, pred, xbuf=None) -> None:
        """D forward on (pred, refined_G): value only (zero gradient to G) but it advances u,v (:361)."""
        B = self.B
        xbuf = self.x0[B:2 * B] if xbuf is None else xbuf          # x0[:B] stays G's input (down1's wgrad operand)
        self.sn.iterate(0, True, zero=self.wgan_mean, defer_finish=self._sn_defer and self._d_dirty)   # (the closing launch clears the mean the head conv adds to)
        self._prep_d()
        ops.pack_pair(pred, self._refined_g, xbuf)
        self._d_forward(B, lambda l: self.sn.isig[l, 0:1], B, x=xbuf, means=self.wgan_mean, groups=1)   # loss_WGAN_G = -mean (:362)

    def g_main(self, pred, delta_true, pred_box, refine_fn, masks) -> None:
        """The generator step's forward, loss and backward (:345-366, without the value-only critic forward).  Two halves so
        that a data-parallel schedule can put one under each critic all-reduce: neither depends on the critic."""
        self.g_main_a(pred, delta_true, pred_box, refine_fn, masks)
        self.g_main_b()

    def g_main_a(self, pred, delta_true, pred_box, refine_fn, masks) -> None:
        """forward (unless the iteration's batched forward already ran), EIoU + its gradient, the re-crop for the critic's
        value forward, and the backward through the head and the up path"""
        B, S = self.B, self.S
        if not self._gall_valid:           # (else: forward done with the iteration's batch; group n_critic = these very buffers)
            self._prep_g()
            ops.pack_pair(pred, None, self.x0[:B])
            self._set_masks(masks, 1)
            self._g_forward(self.x0[:B], True)                                         # :348
        # (lambda_iou only scales the gradient g_gdelta, not the accumulated loss: the generator's loss scale rides on it)
        ops.eiou_fwd_bwd(pred_box, self.g_delta, delta_true, self.lambda_iou * self.loss_scale_g, self.g_gdelta, self.g_cal,
                         self.eiou_acc)                                                # :351-355
        self.delta_pred = self.g_delta                  # (alias: valid until the next generator forward; iteration() clones it)
        if self.delta_log is not None:
            self._log_delta.run()
        self._refined_g = refine_fn(self.delta_pred, self.c)                           # :358-360
        # ---- backward of lambda_iou * EIoU through G (:365-366)
        if not self.G.grads_zero:
            self.G.g.zero_()
        self.G.grads_zero = False
        if self.gen is not None:           # (the simple generator's backward is one piece: g_main_b)
            return
        gW = self.G.gviews
        ops.head_bwd(self.g_gdelta, self.g_traw, self.g_pooled, self.G.views["fc_delta.1.weight"], self.delta_scale,
                     B, S * S, gW["fc_delta.1.weight"], gW["fc_delta.1.bias"], self.g_dab)
        ins = [self.g_d4, self.g_cat1, self.g_cat2, self.g_cat3]                       # inputs of up1..up4
        dcat = [self.g_dd4, self.g_dcat1, self.g_dcat2, self.g_dcat3]                  # grads wrt those inputs
        for k in (3, 2, 1, 0):
            cint, coutt = G_UP[k]
            key = f"up{k + 1}.model.0.weight" if k < 3 else "up4.0.weight"
            if k == 3:
                # (after a fused up4 forward the statistics of this backward -- sum of ReLU' and sum of relu(xhat) per (n, c) --
                #  are its `cnt` output and the head's pooled mean x H*W: maps of > 256 pixels skip their pass over z)
                pre = self._up4_presums and S * S > 256 and os.environ.get("GCSSL_UP4_PRESUM", "1") != "0"      # (A/B knob)
                ops.in_act_bwd(self.g_zu[3], self.g_umean[3], self.g_urstd[3], self.g_dzu[3], coutt, RELU,
                               da_bcast=self.g_dab, ws=self.ws_g, presum_cnt=self.g_ucnt if pre else None,
                               presum_pos=self.g_pooled if pre else None, presum_pos_scale=float(S * S), sat=self.sat_g)
            else:
                ops.in_act_bwd(self.g_zu[k], self.g_umean[k], self.g_urstd[k], self.g_dzu[k], coutt, RELU,
                               da=dcat[k + 1][..., :coutt], mask=self.g_masks[k + 1] if k < 2 else None, ws=self.ws_g, sat=self.sat_g)
            fl = conv_flops(B, S >> (3 - k), coutt, cint)

            def up_wgrad(k=k, cint=cint, coutt=coutt, fl=fl):
                self._conv(f"G.up{k + 1}.wgrad", fl, ops.conv_wgrad, self.g_dzu[k], ins[k], self.g_slab_u[k], coutt,
                           cint)                                                        # roles swapped (ConvTranspose)
            if not (self._wgrad_batch_on and k >= 1):              # (up2-up4: one batched launch behind the chain, as the critic's)
                self._on_side(up_wgrad)
            ns4, st4 = self._split("fwd", self.g_dd4, B, S >> 3, coutt, cint, grad=True) if k == 0 else (1, 0)
            self._conv(f"G.up{k + 1}.dgrad", fl, ops.conv_fwd, self.g_dzu[k], self.gu_wf[k], dcat[k], coutt, cint,
                       split_stride=st4)
        self._g_dd4_slabs = (ns4, st4)
        if self._wgrad_batch_on:
            if self._wgrad_gu is None:
                lay = [(self.g_dzu[k], ins[k], self.g_slab_u[k], G_UP[k][1], G_UP[k][0]) for k in (3, 2, 1)]
                self._wgrad_gu = ops.WgradBatch(lay)
                es = 4 if self.code == _lib.F32 else 2
                self._wgrad_gu_bytes = (sum((x.numel() + dy.numel() + sl.numel() // sl.shape[0]) * es for x, dy, sl, _, _ in lay),
                                        sum(x.numel() * x.element_size() + dy.numel() * dy.element_size() + sl.numel() * 4 for x, dy, sl, _, _ in lay))
            self._conv("G.up2-4.wgrad", sum(conv_flops(B, S >> (3 - k), G_UP[k][1], G_UP[k][0]) for k in (1, 2, 3)), ops.conv_wgrad_batch,
                       self._wgrad_gu, _bytes=self._wgrad_gu_bytes)
        self._join_side()

    def g_main_b(self) -> None:
        """backward through the down path, then all eight weight gradients reduced in one launch"""
        if self.gen is not None:
            self.gen.backward(self.g_gdelta)
            return
        B, S = self.B, self.S
        ns4, st4 = self._g_dd4_slabs
        d_act = [self.g_cat3[..., 64:], self.g_cat2[..., 128:], self.g_cat1[..., 256:]]            # d1, d2, d3
        dskip = [self.g_dcat3[..., 64:], self.g_dcat2[..., 128:], self.g_dcat1[..., 256:]]
        for k in (3, 2, 1, 0):
            cin, cout = G_DOWN[k]
            cp = _pad8(cin)
            if k == 3:
                ops.in_act_bwd(self._g_zsrc[3], self.g_dmean[3], self.g_drstd[3], self.g_dzd[3], 512, LRELU,
                               da=self.g_dd4, mask=self.g_masks[0], ws=self.ws_g, da_nslab=ns4, da_slab_stride=st4, sat=self.sat_g)
            elif k > 0:
                ops.in_act_bwd(self._g_zsrc[k], self.g_dmean[k], self.g_drstd[k], self.g_dzd[k], cout, LRELU,
                               da=self.g_dd[k + 1], da2=dskip[k], ws=self.ws_g, da_nslab=nsd, da_slab_stride=std, sat=self.sat_g)
            else:
                ops.act_bwd(self.g_dd[1], d_act[0], self.g_dzd[0], 64, da2=dskip[0], sat=self.sat_g)
            # (down1's input: the generator's own packed copy when the batched forward ran -- x0[:B] belongs to the critic steps,
            #  which may be re-packing it while this branch runs beside them)
            xin = (self.gfa.x8[self.c * B:] if self._gall_valid else self.x0[:B]) if k == 0 else d_act[k - 1]
            fl = conv_flops(B, S >> k, cin, cout)

            def down_wgrad(k=k, cout=cout, cp=cp, xin=xin, fl=fl):
                self._conv(f"G.down{k + 1}.wgrad", fl, ops.conv_wgrad, xin, self.g_dzd[k], self.g_slab_d[k], cp, cout)
            self._on_side(down_wgrad)
            if k > 1:
                nsd, std = self._split("dgrad", self.g_dd[k], B, S >> k, cin, cout, grad=True)
            else:
                nsd, std = 1, 0                                   # g_dd[1] feeds the norm-less act_bwd: atomic form
            if k > 0:
                self._conv(f"G.down{k + 1}.dgrad", fl, ops.conv_dgrad, self.g_dzd[k], self.gd_wt[k], self.g_dd[k], cin, cout,
                           split_stride=std)
        self._join_side()
        self._reduce_batches()[1].run()                           # all eight weight gradients, one launch

    # ------------------------------------------------------------------------------------------ iteration
    def run_iteration(self, pred, gt, delta_true, pred_box, refine_fn, alphas=None, masks=None, on_critic=None,
                      next_forward: bool = False, next_pred=None):
        """Enqueue one full iteration (n_critic D steps + 1 G step); no host sync, nothing read back.

        next_forward (GraphedIteration's pipelined single-GPU form, device-drawn masks only): the iteration's batched generator
        forward has ALREADY run (by the previous call, or a prologue); this call instead ends its generator branch with the
        NEXT iteration's batched forward -- right behind the generator update, beside the rest of the critic's work -- on
        `next_pred`, the next iteration's input (default: this one's, a fixed batch)."""
        branch = self.overlap_g and self.batch_g and self.probe is None and self.allreduce is None
        sn_early = branch and self.overlap_g >= 2
        if sn_early:
            # (GCSSL_OVERLAP_G=2) the first critic step's spectral-norm chain + weight re-pack (8 tiny launches that depend on
            # nothing of this iteration) beside the batched generator forward; joined in front of the first critic forward
            ev0 = torch.cuda.Event()
            ev0.record(torch.cuda.current_stream())
            self.side_sn.wait_event(ev0)
            with torch.cuda.stream(self.side_sn):
                self._sn_and_prep()
        if next_forward:
            assert branch and self._gall_valid and masks is None, "pipelined form: single GPU, batched forward already done"
        elif self.batch_g:
            self.g_forward_all(pred, masks)
        if branch:
            # the generator step's gradient work (everything of :345-366 but the value-only critic forward) as a parallel
            # branch: it starts behind the batched forward and is joined in front of the generator update
            main = torch.cuda.current_stream()
            ev = torch.cuda.Event()
            ev.record(main)
            self.side_g.wait_event(ev)
            with torch.cuda.stream(self.side_g):
                self._in_g_branch = True
                try:
                    self.g_main(pred, delta_true, pred_box, refine_fn, None if masks is None else masks[self.c])
                finally:
                    self._in_g_branch = False
        ev_pre = None
        for k in range(self.c):
            if branch:
                self.d_pre(pred, gt, refine_fn, k, None if alphas is None else alphas[k], None if masks is None else masks[k])
                if k == self.c - 1:                               # the critic steps' last read of the generator's step counter
                    ev_pre = torch.cuda.Event()                   # (alpha is drawn keyed by it: d_pre, pack_fake_interp)
                    ev_pre.record(torch.cuda.current_stream())
                if sn_early and k == 0:
                    evs = torch.cuda.Event()
                    evs.record(self.side_sn)
                    torch.cuda.current_stream().wait_event(evs)
                self.d_main(sn_done=sn_early and k == 0)
                self.d_update()
            else:
                self.d_step(pred, gt, refine_fn, k, None if alphas is None else alphas[k],
                            None if masks is None else masks[k])
            if on_critic is not None:
                on_critic(k)
        if branch:
            # the generator's clip + Adam goes to the branch too: in STREAM order it follows the backward (it needs nothing
            # from the critic) -- but not before the last d_pre, which reads the step counter this update advances; in host
            # order it comes after the critic steps, whose d_pre still read the host flags of the batched forward it resets
            with torch.cuda.stream(self.side_g):
                ev_gm = torch.cuda.Event()
                ev_gm.record(self.side_g)                         # the backward (and with it the re-crop for g_critic) is enqueued
                self.side_g.wait_event(ev_pre)
                self.g_update()
                if next_forward:
                    # software pipelining across iterations: G's weights are final for this iteration, the critic steps have
                    # taken their deltas (ev_pre) and the backward has read the activations, so the NEXT iteration's batched
                    # forward can overwrite them now -- beside the last critic step's gradient work and the value-only forward
                    self.g_forward_all(pred if next_pred is None else next_pred, None)
            if next_forward:
                torch.cuda.current_stream().wait_event(ev_gm)
            else:
                ev = torch.cuda.Event()
                ev.record(self.side_g)
                torch.cuda.current_stream().wait_event(ev)
            self.g_critic(pred)                                   # :361-362, with the critic's final weights of the iteration
            if next_forward:                                      # (every stream of a capture has to be joined before it ends)
                ev = torch.cuda.Event()
                ev.record(self.side_g)
                torch.cuda.current_stream().wait_event(ev)
        else:
            self.g_step(pred, delta_true, pred_box, refine_fn, None if masks is None else masks[self.c])

    def iteration(self, pred, gt, delta_true, pred_box, refine_fn, alphas=None, masks=None) -> dict:
        """run_iteration + the scalars the reference logs (.item() syncs at :335-337,372-374)."""
        log = {"d_loss": [], "gp": [], "wd": [], "d_grad_norm": [], "real": [], "fake": [], "d_interp": []}

        def grab(_k):
            m = self.means.tolist()
            gp = float(self.gp_sum)
            wd = m[0] - m[1]
            log["wd"].append(wd); log["gp"].append(gp); log["d_loss"].append(-wd + self.lambda_gp * gp)
            log["d_grad_norm"].append(float(self.D.state[2]))
            B = self.B
            log["real"].append(self.d_out[:B].clone()); log["fake"].append(self.d_out[B:2 * B].clone())
            log["d_interp"].append(self.d_out[2 * B:].clone())
        self.run_iteration(pred, gt, delta_true, pred_box, refine_fn, alphas, masks, on_critic=grab)
        loss_iou = 1.0 + float(self.eiou_acc)
        loss_wgan = -float(self.wgan_mean)
        log.update(loss_iou=loss_iou, loss_wgan=loss_wgan, loss_g=self.lambda_iou * loss_iou + loss_wgan,
                   g_grad_norm=float(self.G.state[2]), delta_pred=self.delta_pred.clone(),
                   calibrated=self.g_cal.clone(), fake_for_g=self.d_out[:self.B].clone())
        return log

    def saturations(self) -> dict:
        """fp16 gradient stores clipped so far (host sync): {"critic": n, "generator": n}"""
        c, g = self.sat.tolist()
        return {"critic": int(c), "generator": int(g)}

    def set_lr(self, lr_g: Optional[float] = None, lr_d: Optional[float] = None) -> None:
        """Learning rates for the following updates (an LR scheduler's step): written into the optimisers' device-side
        state blocks, so captured graphs pick them up without re-capture."""
        if lr_g is not None:
            self.G.state[7] = float(lr_g)
        if lr_d is not None:
            self.D.state[7] = float(lr_d)

    # ------------------------------------------------------------------------------------------ state access
    def state_dicts(self):
        """(generator_state_dict, discriminator_state_dict) keyed like the reference."""
        g = {k: v.detach().clone() for k, v in self.G.views.items()}
        d = {k: v.detach().clone() for k, v in self.D.views.items()}
        if self.spectral_norm:
            for l, i in enumerate(D_IDX):
                d[f"model.{i}.weight_u"] = self.u[l].clone()
                d[f"model.{i}.weight_v"] = self.v[l].clone()
        return g, d


class GraphedIteration:
    """One training iteration as hipGraphs: the launch-bound kernel sequences of each critic step and of the generator
    step are captured once (torch.cuda.CUDAGraph == hipGraph on ROCm; our ctypes launches go to the capturing stream)
    and replayed.  Single-GPU runs are ONE graph per iteration.  With data parallelism the iteration is cut where a
    collective's result is needed: only the all-reduce launches and the stream waits sit between the segments, the
    clip+Adam updates are captured at the head of the segment that consumes them.  alpha and dropout masks are drawn on the
    device inside the graph (the counter-based generators keyed by the device-side step count), so every replay differs.

    Inputs.  The captured launches read the four tensors given here (`pred, gt, delta_true, pred_box`: static device buffers).
    ``replay()`` with no arguments re-runs the iteration on whatever they hold -- bench.py's fixed synthetic batch.  A training
    loop feeds a new batch per iteration with ``replay(batch=(pred, gt, delta_true, pred_box), next_pred=<pred of the NEXT
    batch>)``: `batch` is copied into the static buffers, and `next_pred` into a second static buffer that the PIPELINED forms'
    trailing generator forward reads (those forms end replay i with the batched generator forward of iteration i + 1, so they
    need that input one replay early).  A replay that was not told its successor's input -- or one that follows an eager
    `run_iteration` / a load of new weights -- cannot be pipelined into: the next replay then first re-runs the forward on its own
    batch (the prologue graph), which is correct and costs one generator forward.  GCSSL_CHECK_STAGING=1 verifies (host sync)
    that a batch's `pred` is what the previous replay was given as `next_pred`."""

    def __init__(self, eng: "StepEngine", pred, gt, delta_true, pred_box, refine_fn, batch_g_critic: bool = False):
        """batch_g_critic (two-stream form only): iteration i's value-only critic forward (cgan/cgan_train_enhanced.py:361-362) runs
        as a fourth group of iteration i+1's first critic forward (StepEngine.d_main(with_g=True)) instead of as a segment of small
        launches at the end of iteration i -- same arithmetic, same order of power iterations; `loss_wgan` of iteration i (wgan_mean)
        is then available after replay i+1, and after the LAST replay `finish()` runs the one forward that is still owed."""
        self.eng = eng
        self.inputs = (pred, gt, delta_true, pred_box)
        self.batch_g = False
        self._g_owed = False
        self.pred_next = pred.clone()                              # what the pipelined forms' trailing generator forward reads
        self._staged = False                                       # did the previous replay get its successor's input?
        self._primed, self.pipelined = False, False
        self._check_staging = os.environ.get("GCSSL_CHECK_STAGING", "0") != "0"
        self.fused_update = eng.allreduce is None
        # A capture must not depend on host-side state that differs between capture time and replay time.  With
        # keep_clipped_grads the updates leave the CLIPPED gradient in the bucket (clip_grad_norm_ semantics), so every
        # backward has to start with its zero fill -- also the first one of a fresh engine, whose buckets happen to be zero:
        # mark them dirty so that the fill is part of the graph.  (Without keep_clipped_grads the update re-zeroes the bucket
        # it consumed, on the device, in every replay: the flag is then true at capture time and at every replay.)
        if eng.keep_clipped_grads:
            eng.D.grads_zero = eng.G.grads_zero = False
        self.d_graphs, self.g_graph = [], None
        pool = None

        def capture(fn):
            nonlocal pool
            g = torch.cuda.CUDAGraph()
            # thread_local: with data parallelism RCCL's watchdog thread polls events while we capture; only THIS thread's
            # calls belong to the capture
            with torch.cuda.graph(g, pool=pool, capture_error_mode="thread_local"):
                fn()
            pool = g.pool()
            return g

        if self.fused_update and eng.overlap_g and eng.batch_g and os.environ.get("GCSSL_ONE_GRAPH", "1") != "0":
            # single GPU, ONE graph per iteration with the generator step's gradient work as a parallel branch beside the critic
            # steps (StepEngine.run_iteration: one fork behind the batched generator forward, one join in front of the updates)
            # GCSSL_PIPELINE (default on): software pipelining across iterations -- the replayed graph is [critic steps i,
            # value-only forward i] beside [generator backward i, generator update i, batched generator forward i+1]; a prologue
            # graph runs the very first forward.  Same launches per iteration, same dependency order (the forward of i+1 needs
            # exactly G's weights after update i and iteration i+1's input, which is this static batch).
            self.pipelined = os.environ.get("GCSSL_PIPELINE", "1") != "0"
            self._primed = False

            def whole():
                eng._d_dirty = True                               # (capture-time host state must not skip the re-packs)
                if not self.pipelined:
                    eng._g_dirty = True
                eng.run_iteration(pred, gt, delta_true, pred_box, refine_fn, next_forward=self.pipelined, next_pred=self.pred_next)
            # GCSSL_TWO_STREAM (default on, pipelined form only): the same launches in the same dependency order as FOUR LINEAR
            # graphs on two streams instead of one graph with a parallel branch --
            #   this stream:  [Ca: critic steps 0..c-2, d_pre of the last]  [Cb: last critic step, value-only forward]
            #   side stream:  [Ga: generator backward]                      [Gb: generator update, batched forward i+1]
            #   Ca(i) after Gb(i-1);  Gb(i) after Ca(i) (the last d_pre's reads);  Cb(i) after Ga(i) (the re-crop);
            #   Ga(i+1) after Cb(i) (it rewrites that re-crop)
            # The runtime's executor for a graph WITH branches left the critic's chain idle for ~240 us at the head of every replay
            # and ~190 us at its tail (profiles/round3_timeline_one_graph.txt: 77 % of the period with ONE kernel resident); linear
            # graphs replay as pre-built packet streams, and the generator's chain now runs a segment ahead of the critic's.
            self.two_stream = self.pipelined and os.environ.get("GCSSL_TWO_STREAM", "1") != "0"
            if self.two_stream:
                c = eng.c
                self.side = _side_stream(eng.dev, "gen")
                eng._g_dirty = True
                pool_main, pool = pool, None                       # the generator's graphs replay beside the critic's: own pool
                self.prologue = capture(lambda: eng.g_forward_all(pred, None))
                eng._in_g_branch = True                            # (no forks inside a segment: every graph stays linear)
                try:
                    self.g_a = capture(lambda: eng.g_main(pred, delta_true, pred_box, refine_fn, None))
                    pool_g, pool = pool, pool_main

                    # GCSSL_HEAD_SPLIT=1 (A/B knob, default OFF): the first and the last critic step's spectral-norm chain + weight
                    # re-pack -- eight 7-us launches, each depending on the one before -- as graphs of their own (c_a0, c_b0), with
                    # the generator's segments started BEHIND them (g_a after c_a0, g_b after c_b0).  The idea: started together,
                    # the generator's chain of one-workgroup-per-CU launches gets the CUs first at every launch boundary and the
                    # critic's tiny launches wait for them (92 / 129 us for a chain that takes 39 us alone, plus a 220-us idle
                    # head: tools/graph_gaps.sh), while the generator's chain has ~0.5 ms of slack.  MEASURED (same box): bf16
                    # 135.06k vs 135.05-135.59k without, fp16x3 66.6k vs 68.2k -- what the critic's heads gain, the co-residency
                    # of the big launches that now coincide loses.  The chip's throughput is shared, not idle.
                    self.head_split = os.environ.get("GCSSL_HEAD_SPLIT", "0") != "0"

                    def seg_a(sn_done=False):
                        eng._d_dirty = True
                        for k in range(c):
                            if not (sn_done and k == 0):
                                eng.d_pre(pred, gt, refine_fn, k, None, None)
                            if k < c - 1:
                                eng.d_main(sn_done=sn_done and k == 0); eng.d_update()
                    self.batch_g = bool(batch_g_critic) and c >= 2 and eng.gbatch_ok() and os.environ.get("GCSSL_GCRITIC_BATCH", "1") != "0"
                    if self.batch_g:
                        def seg_a_g():
                            eng._d_dirty = True
                            for k in range(c):
                                eng.d_pre(pred, gt, refine_fn, k, None, None)
                                if k < c - 1:
                                    eng.d_main(with_g=(k == 0)); eng.d_update()
                        self.c_a_g = capture(seg_a_g)             # ... with the previous iteration's value-only forward as a fourth group

                        def seg_a_0():                            # (first replay / after finish(): nothing owed; wgan_mean reads 0)
                            eng._d_dirty = True
                            for k in range(c):
                                eng.d_pre(pred, gt, refine_fn, k, None, None)
                                if k < c - 1:
                                    eng.d_main(zero_wgan=(k == 0)); eng.d_update()
                        self.c_a = capture(seg_a_0)
                        # the last critic step, then only the PACK of this iteration's value-only forward: the next replay runs it
                        self.c_b = capture(lambda: (eng.d_main(), eng.d_update(), ops.pack_pair(pred, eng._refined_g, eng.x4[3 * eng.B:])))
                        self.head_split = False
                    elif self.head_split and c >= 2:
                        def head_a():
                            eng._d_dirty = True
                            eng.d_pre(pred, gt, refine_fn, 0, None, None)
                            eng._sn_and_prep()
                        self.c_a0 = capture(head_a)
                        self.c_a = capture(lambda: seg_a(True))
                        self.c_b0 = capture(lambda: (setattr(eng, "_d_dirty", True), eng._sn_and_prep()))
                        self.c_b = capture(lambda: (eng.d_main(sn_done=True), eng.d_update(), eng.g_critic(pred)))
                    else:
                        self.head_split = False
                        self.c_a = capture(seg_a)
                        self.c_b = capture(lambda: (eng.d_main(), eng.d_update(), eng.g_critic(pred)))
                    pool = pool_g
                    self.g_b = capture(lambda: (eng.g_update(), eng.g_forward_all(self.pred_next, None)))
                finally:
                    eng._in_g_branch = False
                self.ev_cb = self.ev_gb = None
                return
            if self.pipelined:
                eng._g_dirty = True
                self.prologue = capture(lambda: eng.g_forward_all(pred, None))
            self.graphs = [capture(whole)]
            return
        if self.fused_update:
            # single GPU: the whole iteration (critic steps, generator step, all three updates) is ONE graph
            # (GCSSL_ONE_GRAPH=0: one graph per step, for A/B runs)
            steps = []
            for k in range(eng.c):
                steps.append(lambda k=k: (setattr(eng, "_d_dirty", True), setattr(eng, "_g_dirty", k == 0),
                                          eng.g_forward_all(pred, None) if (k == 0 and eng.batch_g) else None,
                                          eng.d_compute(pred, gt, refine_fn, k, None, None), eng.d_update()))
            steps.append(lambda: (setattr(eng, "_d_dirty", True), setattr(eng, "_g_dirty", False),
                                  eng.g_compute(pred, delta_true, pred_box, refine_fn, None), eng.g_update()))
            if os.environ.get("GCSSL_ONE_GRAPH", "1") != "0":
                self.graphs = [capture(lambda: [f() for f in steps])]
            else:
                self.graphs = [capture(f) for f in steps]
            return
        # data parallel: every all-reduce runs beside launches that do not need its result --
        #   critic step k's gradient  ||  d_pre of step k+1 (G's no-grad forward, re-crop, packing) or, after the last
        #                                 critic step, g_main (G's forward + backward)
        #   the generator's gradient  ||  the last critic update + g_critic (the value-only critic forward of the G step)
        # and every optimiser update is captured at the head of the segment that needs its result (the stream only WAITS
        # for the collective between two graph launches; the 1/world factor is baked into the fused clip+Adam node):
        #   [pre0 main0] AR(D) [pre1] wait [updD main1] AR(D) ... [g_main] AR(G) wait(D) [updD g_critic] wait(G) [updG]
        if getattr(eng.allreduce, "start", None) is None or getattr(eng.allreduce, "world", None) is None:
            raise ValueError("GraphedIteration's data-parallel form needs a dist.GradAverager-like hook (start/finish/world)")
        gs = eng.dp_grad_scale()
        eng._g_dirty, eng._d_dirty = True, True
        # Round 3: with the batched generator forward the generator step's gradient work is its own graph, replayed on a SECOND
        # stream beside everything the critic does (it depends on nothing of it: StepEngine.run_iteration does the same inside
        # the single-GPU graph).  The critic's exchanges then no longer need that work to hide behind: while the compute stream
        # waits for an all-reduce, the generator's kernels run --
        #   main: [Gfwd_all] [pre0 main0] AR(D) [pre1] wait [updD main1] AR(D) . AR(G) wait(D) [updD g_critic] wait(G) [updG]
        #   side:            [g_main ...................]                   join^
        # GCSSL_DP_BRANCH=0: the round-2 schedule (generator halves serial under the critic's all-reduces).
        self.dp_branch = eng.batch_g and os.environ.get("GCSSL_DP_BRANCH", "1") != "0"
        # GCSSL_DP_PIPELINE=1 (default OFF since round 4: it issues collectives on TWO RCCL communicators from two streams with
        # nothing ordering them on the device -- a documented hang hazard that no multi-rank RCCL run has ever exercised; the
        # default is the single-communicator dp_branch schedule below): the single-GPU two-stream schedule with the exchanges in it -- the generator's chain
        # (backward, all-reduce of ITS gradient on a communicator of its own, update, the NEXT iteration's batched forward) runs
        # whole on the second stream beside the critic's, so no generator launch is left on the critic's stream and the 25-MB
        # exchange never queues in front of a critic exchange:
        #   main: [d_pre0 d_main0] AR(D) [d_pre1] wait [updD d_main1] AR(D) .............. wait(D) [updD g_critic]
        #   side: [g_main] AR'(G) ........................ wait(last d_pre, G) [updG  Gfwd i+1]          (AR' = second process group)
        # The collectives of a group are issued in the same program order on every rank.
        self.dp_pipeline = (self.dp_branch and os.environ.get("GCSSL_DP_PIPELINE", "0") != "0" and
                            getattr(eng.allreduce, "group", "x") != "x" and torch.distributed.is_initialized())
        if self.dp_pipeline:
            c = eng.c
            self.side = _side_stream(eng.dev, "gen")
            self.g_avg = type(eng.allreduce)(group=torch.distributed.new_group())    # (collective: every rank builds its engine here)
            self._primed, self.pipelined = False, True
            pool_main, pool = pool, None                                   # the generator's graphs replay beside the critic's: own pool
            self.prologue = capture(lambda: eng.g_forward_all(pred, None))
            eng._g_dirty = False
            eng._in_g_branch = True                                        # (no forks inside a segment: every graph stays linear)
            try:
                self.g_a = capture(lambda: eng.g_main(pred, delta_true, pred_box, refine_fn, None))
                pool_g, pool = pool, pool_main
                self.first = capture(lambda: (eng.d_pre(pred, gt, refine_fn, 0, None, None), eng.d_main()))
                self.pre, self.upd_main = [None], [None]
                for k in range(1, c):
                    self.pre.append(capture(lambda k=k: eng.d_pre(pred, gt, refine_fn, k, None, None)))
                    self.upd_main.append(capture(lambda: (eng.d_update(gs), eng.d_main())))
                self.upd_crit = capture(lambda: (eng.d_update(gs), eng.g_critic(pred)))
                pool = pool_g
                self.g_b = capture(lambda: (eng.g_update(gs), eng.g_forward_all(self.pred_next, None)))
            finally:
                eng._in_g_branch = False
            return
        if self.dp_branch:
            self.side = _side_stream(eng.dev, "gen", int(os.environ.get("GCSSL_SIDE_PRIO", "0")))
            self.gfwd = capture(lambda: eng.g_forward_all(pred, None))
            eng._g_dirty = False
            self.first = capture(lambda: (eng.d_pre(pred, gt, refine_fn, 0, None, None), eng.d_main()))
            self.pre, self.upd_main = [None], [None]
            for k in range(1, eng.c):
                self.pre.append(capture(lambda k=k: eng.d_pre(pred, gt, refine_fn, k, None, None)))
                self.upd_main.append(capture(lambda: (eng.d_update(gs), eng.d_main())))
            pool_main, pool = pool, None                           # the branch replays concurrently: its own memory pool
            self.g_branch = capture(lambda: eng.g_main(pred, delta_true, pred_box, refine_fn, None))
            pool = pool_main
            self.upd_crit = capture(lambda: (eng.d_update(gs), eng.g_critic(pred)))
            self.upd_g = capture(lambda: eng.g_update(gs))
            return
        self.first = capture(lambda: (eng.g_forward_all(pred, None) if eng.batch_g else None,
                                      eng.d_pre(pred, gt, refine_fn, 0, None, None), eng.d_main()))
        # the generator step's own work (forward already done, EIoU, backward) does not depend on the critic: with two or
        # more critic steps its first half rides with pre[1] under the first critic all-reduce, its second half under the last
        self.pre, self.upd_main = [None], [None]
        split_g = eng.c >= 2 and eng.batch_g       # (one forward per call would re-pack x0[:B] between two critic steps)
        for k in range(1, eng.c):
            eng._g_dirty = False
            self.pre.append(capture(lambda k=k: (eng.d_pre(pred, gt, refine_fn, k, None, None),
                                                 eng.g_main_a(pred, delta_true, pred_box, refine_fn, None) if (k == 1 and split_g) else None)))
            self.upd_main.append(capture(lambda: (eng.d_update(gs), eng.d_main())))
        eng._g_dirty = False
        self.g_main = capture(eng.g_main_b if split_g else (lambda: eng.g_main(pred, delta_true, pred_box, refine_fn, None)))
        self.upd_crit = capture(lambda: (eng.d_update(gs), eng.g_critic(pred)))
        self.upd_g = capture(lambda: eng.g_update(gs))

    def _stage(self, batch, next_pred) -> None:
        """Copy a new batch / the next batch's generator input into the static buffers (on the caller's stream, which every
        launch of the replay is ordered behind) and decide whether the pending pipelined forward is this iteration's."""
        eng = self.eng
        if batch is not None:
            if self.pipelined and self._staged and self._check_staging and not torch.equal(self.pred_next, batch[0]):
                raise RuntimeError("GraphedIteration.replay: this batch's pred is not what the previous replay got as next_pred")
            for dst, src in zip(self.inputs, batch):
                if src is not dst:
                    dst.copy_(src, non_blocking=True)
        if not self.pipelined:
            return
        # the forward the previous replay left pending is usable iff it ran on THIS iteration's input with the current weights
        # and nothing has overwritten its activations since: the previous replay was told this batch (or both run on the
        # unchanged static buffers), and no eager iteration / generator forward touched the engine in between
        ok = self._primed and eng._gall_valid and (self._staged if batch is not None else not self._staged)
        if next_pred is not None:
            self.pred_next.copy_(next_pred, non_blocking=True)
        elif batch is not None or self._staged:
            self.pred_next.copy_(self.inputs[0], non_blocking=True)   # no successor announced: assume the same batch again
        self._staged = next_pred is not None
        self._primed = ok

    def finish(self) -> None:
        """batch_g_critic: run the value-only critic forward the last replay left owed (no-op otherwise).  Call it after the last
        replay of a run, and before anything else touches the engine (an eager iteration, a checkpoint of u / v, reading wgan_mean)."""
        if self._g_owed:
            self.eng.g_critic(self.inputs[0])
            self._g_owed = False

    def replay(self, batch=None, next_pred=None):
        """One iteration.  batch: optional (pred, gt, delta_true, pred_box) for this iteration; next_pred: the NEXT iteration's
        generator input, for the pipelined forms (class docstring)."""
        eng = self.eng
        self._stage(batch, next_pred)
        if self.fused_update and getattr(self, "two_stream", False):
            main, side = torch.cuda.current_stream(), self.side
            if not self._primed:
                self.prologue.replay()                            # this iteration's batched generator forward (first replay / re-prime)
                self._primed = True
            eng._gall_valid = True                                # (host flag: every replay leaves the next iteration's forward pending)
            if self.head_split:
                self.c_a0.replay()                                # d_pre(0) + the first critic step's spectral-norm chain and re-pack
            ev0 = torch.cuda.Event(); ev0.record(main)            # (Gb(i-1) and Cb(i-1) are ordered in front of this: see the tail)
            side.wait_event(ev0)
            with torch.cuda.stream(side):
                self.g_a.replay()
                ev_ga = torch.cuda.Event(); ev_ga.record(side)
            if self.batch_g and self._g_owed:
                self.c_a_g.replay()                               # (with the previous iteration's value-only forward)
            else:
                self.c_a.replay()
            self._g_owed = self.batch_g
            if self.head_split:
                self.c_b0.replay()                                # the last critic step's chain (Gb starts behind it)
            ev_ca = torch.cuda.Event(); ev_ca.record(main)
            with torch.cuda.stream(side):
                side.wait_event(ev_ca)
                self.g_b.replay()
                ev_gb = torch.cuda.Event(); ev_gb.record(side)
            main.wait_event(ev_ga)
            self.c_b.replay()
            main.wait_event(ev_gb)                                # the caller's stream sees the whole iteration
            return
        if self.fused_update:
            if self.pipelined and not self._primed:
                self.prologue.replay()                            # this iteration's batched generator forward (first replay / re-prime)
                self._primed = True
            if self.pipelined:
                eng._gall_valid = True                            # (host flag: the replay leaves the next iteration's forward pending)
            for g in self.graphs:
                g.replay()
            return
        if getattr(self, "dp_pipeline", False):
            main, side = torch.cuda.current_stream(), self.side
            if not self._primed:
                self.prologue.replay()                            # this iteration's batched generator forward (first replay / re-prime)
                self._primed = True
            eng._gall_valid = True                                # (host flag: every replay leaves the next iteration's forward pending)
            ev0 = torch.cuda.Event(); ev0.record(main)            # (the previous replay ended with main behind its g_b)
            side.wait_event(ev0)
            with torch.cuda.stream(side):
                self.g_a.replay()
                ev_ga = torch.cuda.Event(); ev_ga.record(side)
                hg = self.g_avg.start(eng.G.g)                    # the generator's exchange: its own communicator, behind g_a
            self.first.replay()
            ev_pre = torch.cuda.Event()
            if eng.c == 1:
                ev_pre.record(main)
            h = eng.allreduce_start(eng.D.g)
            for k in range(1, eng.c):
                self.pre[k].replay()
                if k == eng.c - 1:
                    ev_pre.record(main)                           # the iteration's last d_pre (its reads of the deltas and of G's step count)
                eng.allreduce_wait(h, eng.D.g)
                self.upd_main[k].replay()
                h = eng.allreduce_start(eng.D.g)
            with torch.cuda.stream(side):
                side.wait_event(ev_pre)
                self.g_avg.finish(hg, eng.G.g, scale=False)       # (1 / world is baked into the captured update)
                self.g_b.replay()
                ev_gb = torch.cuda.Event(); ev_gb.record(side)
            main.wait_event(ev_ga)                                # the re-crop the value-only forward packs
            eng.allreduce_wait(h, eng.D.g)
            self.upd_crit.replay()
            main.wait_event(ev_gb)                                # the caller's stream sees the whole iteration
            return
        if self.dp_branch:
            main = torch.cuda.current_stream()
            self.gfwd.replay()
            ev = torch.cuda.Event(); ev.record(main)
            self.side.wait_event(ev)
            with torch.cuda.stream(self.side):
                self.g_branch.replay()
                evg = torch.cuda.Event(); evg.record(self.side)
            self.first.replay()
            h = eng.allreduce_start(eng.D.g)
            for k in range(1, eng.c):
                self.pre[k].replay()
                eng.allreduce_wait(h, eng.D.g)
                self.upd_main[k].replay()
                h = eng.allreduce_start(eng.D.g)
            main.wait_event(evg)                                  # the generator's gradient is complete
            hg = eng.allreduce_start(eng.G.g)                     # queued behind the critic's exchange on the backend's stream
            eng.allreduce_wait(h, eng.D.g)
            self.upd_crit.replay()
            eng.allreduce_wait(hg, eng.G.g)
            self.upd_g.replay()
            return
        self.first.replay()
        h = eng.allreduce_start(eng.D.g)
        for k in range(1, eng.c):
            self.pre[k].replay()
            eng.allreduce_wait(h, eng.D.g)
            self.upd_main[k].replay()
            h = eng.allreduce_start(eng.D.g)
        self.g_main.replay()
        hg = eng.allreduce_start(eng.G.g)                         # queued behind the critic's exchange on the backend's stream
        eng.allreduce_wait(h, eng.D.g)
        self.upd_crit.replay()
        eng.allreduce_wait(hg, eng.G.g)
        self.upd_g.replay()
