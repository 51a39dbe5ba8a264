"""GeneratorSimpleRegressor (generator_type "simple": cgan/models.py:147-216, selected at
cgan/cgan_train_enhanced.py:26-31) as an explicit kernel schedule on the step engine's buffers.

    features : 4 x [ Conv3x3(+bias) -> InstanceNorm -> ReLU ] x2 -> MaxPool2d(2,2)     (64, 128, 256, 512 channels)
    regressor: AdaptiveAvgPool2d(1) -> Linear(512,256) ReLU Dropout -> Linear(256,64) ReLU Dropout -> Linear(64,4) -> Tanh
    delta = tanh * delta_scale

The engine owns the optimiser state (one flat fp32 parameter / gradient / moment buffer), the clip+Adam launch, the
hipGraph capture and the data-parallel all-reduce; this class provides ``prep`` (weight re-pack after an update),
``forward(x8, train)`` and ``backward(gdelta)``, which writes every gradient into the flat buffer.  Layout as everywhere
in the engine: NHWC activations in the compute dtype, fp32 pre-norm tensors, fp32 gradients where a norm / pool backward
consumes them, compute-dtype gradients (dz) where an MFMA does.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from . import ops
from .ops import RELU

GS_CONV = [(3, 64), (64, 64), (64, 128), (128, 128), (128, 256), (256, 256), (256, 512), (512, 512)]
GS_CONV_IDX = (0, 3, 7, 10, 14, 17, 21, 24)           # positions of the nn.Conv2d modules in .features
GS_FC_IDX = (2, 5, 8)                                  # positions of the nn.Linear modules in .regressor
GS_PARAM_KEYS = ([k for i in GS_CONV_IDX for k in (f"features.{i}.weight", f"features.{i}.bias")]
                 + [k for i in GS_FC_IDX for k in (f"regressor.{i}.weight", f"regressor.{i}.bias")])   # named_parameters() order


def conv3_flops(n: int, h: int, cin: int, cout: int) -> float:
    """Algorithmic FLOPs (2*MAC) of one 3x3 s1 p1 conv pass over n samples of h x h maps."""
    return 2.0 * n * h * h * cout * 9 * cin


class SFwd:
    """Forward-pass buffers of the simple generator for a batch of n samples."""
    LISTS = ("z", "a", "mean", "rstd", "p", "masks")
    FIELDS = ("feat", "h1", "h2", "x8", "traw", "delta")

    def group(self, g: int, b: int) -> "SFwd":
        sl, v = slice(g * b, (g + 1) * b), SFwd()
        v.n = b
        for f in self.FIELDS:
            setattr(v, f, getattr(self, f)[sl])
        for f in self.LISTS:
            setattr(v, f, [t[sl] for t in getattr(self, f)])
        return v


class SimpleGenerator:
    def __init__(self, eng):
        self.eng = eng
        B, S, T, dev = eng.B, eng.S, eng.T, eng.dev
        f32 = dict(device=dev, dtype=torch.float32)
        self.res = [S >> (j // 2) for j in range(8)]
        self.cinp = [max(8, cin) for cin, _ in GS_CONV]
        # forward buffers: ONE set for all n_critic + 1 generator calls of an iteration (forward_all runs them as one
        # batch, StepEngine.g_forward_all); the generator step's group is what backward and the stand-alone calls use
        NG = getattr(eng, "c", 0) + 1
        n = NG * B
        fa = SFwd()
        fa.n = n
        fa.z, fa.a, fa.mean, fa.rstd = [], [], [], []
        self.wf, self.wt, self.dz, self.da, self.slab, self.ns = ([] for _ in range(6))
        for j, ((cin, cout), r) in enumerate(zip(GS_CONV, self.res)):
            cp = self.cinp[j]
            self.wf.append(torch.empty(cout, ops.conv3_wk(cp), device=dev, dtype=T))
            self.wt.append(torch.empty(cin, ops.conv3_wk(cout), device=dev, dtype=T) if j > 0 else None)
            fa.z.append(torch.empty(n, r, r, cout, **f32))
            fa.a.append(torch.empty(n, r, r, cout, device=dev, dtype=T))
            fa.mean.append(torch.empty(n, cout, **f32))
            fa.rstd.append(torch.empty(n, cout, **f32))
            self.dz.append(torch.empty(B, r, r, cout, device=dev, dtype=T))
            self.da.append(torch.empty(B, r, r, cout, **f32))
            ns = ops.conv3_wgrad_splits(B, r, cp, cout)
            self.ns.append(ns)
            self.slab.append(torch.empty(ns, cout, 16, cp, **f32))
        # pooled outputs of the four blocks and the gradients that arrive at them (blocks 0..2: from the next conv's dgrad)
        fa.p = [torch.empty(n, self.res[2 * b + 1] // 2, self.res[2 * b + 1] // 2, GS_CONV[2 * b + 1][1], device=dev, dtype=T)
                for b in range(4)]
        self.dp = [torch.empty(B, self.res[2 * b + 1] // 2, self.res[2 * b + 1] // 2, GS_CONV[2 * b + 1][1], **f32)
                   for b in range(3)]
        fa.feat = torch.empty(n, 512, **f32)
        fa.h1, fa.h2 = torch.empty(n, 256, **f32), torch.empty(n, 64, **f32)
        self.dp1, self.dp2, self.dp3 = torch.empty(B, 256, **f32), torch.empty(B, 64, **f32), torch.empty(B, 4, **f32)
        self.dfeat = torch.empty(B, 512, **f32)
        self.maskbuf = torch.empty(n * (256 + 64), device=dev, dtype=torch.uint8)       # one launch draws both, for every call
        fa.masks = [self.maskbuf[:n * 256].view(n, 256), self.maskbuf[n * 256:].view(n, 64)]
        fa.x8 = torch.empty(n, S, S, 8, device=dev, dtype=T)
        ga = getattr(eng, "gfa", None)                                # the engine's (n, 4) head outputs, shared with the U-Net path
        fa.traw = ga.traw if ga is not None else eng.g_traw
        fa.delta = ga.delta if ga is not None else eng.g_delta
        self.fa = fa
        self.f = f = fa.group(NG - 1, B)
        self.z, self.a, self.mean, self.rstd, self.p, self.masks = f.z, f.a, f.mean, f.rstd, f.p, f.masks
        self.feat, self.h1, self.h2 = f.feat, f.h1, f.h2
        self.w1t, self.w2t = torch.empty(512, 256, **f32), torch.empty(256, 64, **f32)   # transposed head weights (forward)
        self._prep = self._red = None

    # ---------------------------------------------------------------------------------------------- weights
    def _w(self, j: int):
        i = GS_CONV_IDX[j]
        return f"features.{i}.weight", f"features.{i}.bias"

    def prep(self):
        """fp32 [Cout][Cin][3][3] -> the forward and the rotated-transposed (data gradient) operand packs, one launch."""
        if self._prep is None:
            V = self.eng.G.views
            self._prep = ops.Prep3Batch([(V[self._w(j)[0]], self.wf[j], self.wt[j], cout, cin, self.cinp[j])
                                         for j, (cin, cout) in enumerate(GS_CONV)], self.eng.mma)
        self._prep.run()
        self.w1t.copy_(self.eng.G.views["regressor.2.weight"].t())
        self.w2t.copy_(self.eng.G.views["regressor.5.weight"].t())

    def set_masks(self, masks: Optional[Sequence[torch.Tensor]], phase: int):
        """The two Dropout(0.5) keep-masks (models.py:205,208): given (fixture / parity mode) or drawn on the device with
        the engine's counter-based generator (same keying as the U-Net's masks: StepEngine._set_masks)."""
        if masks is None:
            for j, m in enumerate(self.masks):                         # (views into the all-calls buffer: one launch each)
                ops.dropout_mask_gen(m, self.eng.seed * 131 + phase + 7919 * j, self.eng.G.state)
        else:
            for m, src in zip(self.masks, masks):
                m.copy_(src)

    # ---------------------------------------------------------------------------------------------- forward
    def forward(self, x8: torch.Tensor, train: bool = True) -> torch.Tensor:
        """models.py:213-216 on an NHWC8 input whose first 3 channels are pred (the generator step's group)."""
        self.x8 = x8                                   # (kept: the first conv's weight gradient contracts against it)
        return self._forward(x8, train, self.f)

    def forward_all(self, pred: torch.Tensor, masks=None) -> None:
        """All n_critic + 1 generator calls of an iteration as one batch (see StepEngine.g_forward_all)."""
        eng, fa, B = self.eng, self.fa, self.eng.B
        ng = fa.n // B
        ops.pack_pair(pred, None, fa.x8, reps=ng)          # the same input for every call
        if masks is None:
            ops.dropout_mask_gen(self.maskbuf, eng.seed * 131 + 20, eng.G.state)
        else:
            for g, pair in enumerate(masks):
                for m, src in zip(fa.masks, pair):
                    m[g * B:(g + 1) * B].copy_(src)
        self.x8 = fa.x8[(ng - 1) * B:]
        self._forward(fa.x8, True, fa)

    def _forward(self, x8: torch.Tensor, train: bool, f: SFwd) -> torch.Tensor:
        eng, n, V = self.eng, f.n, self.eng.G.views
        tag = "" if n == eng.B else f"[n={n}]"
        src = x8
        for j, (cin, cout) in enumerate(GS_CONV):
            wk, bk = self._w(j)
            eng._conv(f"GS.c{j + 1}.fwd{tag}", conv3_flops(n, self.res[j], cin, cout), ops.conv3_fwd, src, self.wf[j], f.z[j],
                      self.cinp[j], cout, bias=V[bk])
            ops.in_act_fwd(f.z[j], f.a[j], f.mean[j], f.rstd[j], cout, RELU)
            if j & 1:
                ops.maxpool2_fwd(f.a[j], f.p[j // 2], cout)
                src = f.p[j // 2]
            else:
                src = f.a[j]
        ops.avgpool_fwd(f.p[3], f.feat, 512)
        m1, m2 = f.masks if train else (None, None)
        ops.mlp_head_fwd(f.feat, self.w1t, V["regressor.2.bias"], self.w2t,
                         V["regressor.5.bias"], V["regressor.8.weight"], V["regressor.8.bias"], eng.delta_scale,
                         f.h1, f.h2, f.traw, f.delta, m1=m1, m2=m2)
        return f.delta

    # ---------------------------------------------------------------------------------------------- backward
    def backward(self, gdelta: torch.Tensor):
        """Gradient of the generator loss wrt every parameter, given d loss / d delta (the EIoU kernel's output), written
        into the engine's flat gradient buffer (plain stores; the conv biases in front of InstanceNorm keep the exact
        zero gradient they have analytically)."""
        eng, B = self.eng, self.eng.B
        V, gW = eng.G.views, eng.G.gviews
        ops.mlp_head_bwd(gdelta, eng.g_traw, self.h1, self.h2, self.feat, V["regressor.2.weight"], V["regressor.5.weight"],
                         V["regressor.8.weight"], eng.delta_scale, True, self.dp1, self.dp2, self.dp3, self.dfeat,
                         gW["regressor.2.weight"], gW["regressor.2.bias"], gW["regressor.5.weight"], gW["regressor.5.bias"],
                         gW["regressor.8.weight"], gW["regressor.8.bias"])
        for j in range(7, -1, -1):
            cin, cout = GS_CONV[j]
            r = self.res[j]
            if j == 7:        # AdaptiveAvgPool2d(1) backward folded into the pool backward: dfeat / (h*w) at every pooled pixel
                ops.maxpool2_bwd(self.a[7], self.dfeat, self.da[7], cout, bcast_scale=1.0 / ((r // 2) * (r // 2)))
            elif j & 1:
                ops.maxpool2_bwd(self.a[j], self.dp[j // 2], self.da[j], cout)
            # (no dbias: the conv bias sits in front of InstanceNorm, its true gradient is exactly zero -- sum_p dz = 0 --
            #  and the engine leaves it at the zero the gradient buffer was cleared to; autograd's value is rounding noise)
            ops.in_act_bwd(self.z[j], self.mean[j], self.rstd[j], self.dz[j], cout, RELU, da=self.da[j], ws=eng.ws_g)
            src = self.x8 if j == 0 else (self.p[j // 2 - 1] if not (j & 1) else self.a[j - 1])
            fl = conv3_flops(B, r, cin, cout)
            eng._conv(f"GS.c{j + 1}.wgrad", fl, ops.conv3_wgrad, src, self.dz[j], self.slab[j], self.cinp[j], cout)
            if j > 0:         # gradient wrt this conv's input: the previous ReLU output (odd j) or the previous block's pool
                dst = self.da[j - 1] if (j & 1) else self.dp[j // 2 - 1]
                eng._conv(f"GS.c{j + 1}.dgrad", fl, ops.conv3_fwd, self.dz[j], self.wt[j], dst, cout, cin)
        if self._red is None:
            self._red = ops.Reduce3Batch([dict(slab=self.slab[j], nsplit=self.ns[j], dw=gW[self._w(j)[0]], cout=cout,
                                               cin=self.cinp[j], cin_real=cin) for j, (cin, cout) in enumerate(GS_CONV)])
        self._red.run()
