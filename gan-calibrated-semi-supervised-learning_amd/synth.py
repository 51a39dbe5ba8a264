"""Deterministic synthetic weights and step inputs (host side, numpy only).

Everything here is produced by an exact integer counter hash (splitmix64) followed by
exactly-rounded float64 arithmetic (no libm transcendental), so the same call returns
bit-identical arrays in the dev container and on the GPU box.  That is what lets the
golden fixtures under ``tests/golden/`` hold only inputs/outputs: the 9 M weights are
regenerated from a seed instead of being committed.

Shapes/ranges follow the step input contract of the reference
(``cgan/dataset.py:222-236`` as summarised in SURVEY.md §8a row I and §8d):
``pred/gt/refined`` patches ~ U[-1,1] (B,3,S,S); ``pred_box`` cx,cy~U[.3,.7], w,h~U[.1,.5];
``delta_true`` ~ N(0,0.1^2) clipped; ``alpha`` ~ U[0,1); dropout keep-masks ~ Bernoulli(.5).
"""
from __future__ import annotations

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def _stream_id(name: str, seed: int) -> int:
    """Stable 64-bit id of a named stream (FNV-1a over the name, mixed with the seed)."""
    h = 0xCBF29CE484222325
    for ch in name.encode("utf-8"):
        h = ((h ^ ch) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return (h ^ (seed * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF


def uniform01(name: str, seed: int, n: int, lane: int = 0) -> np.ndarray:
    """n float64 values in [0,1) with 24 random bits each (exactly representable in fp32)."""
    with np.errstate(over="ignore"):
        base = np.uint64(_stream_id(name, seed)) + np.uint64(lane) * np.uint64(0xD1B54A32D192ED03)
        idx = np.arange(n, dtype=np.uint64)
        h = _splitmix64(_splitmix64(idx ^ base) + base)
    return (h >> np.uint64(40)).astype(np.float64) * (1.0 / 16777216.0)


def uniform(name: str, seed: int, shape, lo: float = 0.0, hi: float = 1.0) -> np.ndarray:
    n = int(np.prod(shape))
    return (lo + (hi - lo) * uniform01(name, seed, n)).astype(np.float32).reshape(shape)


def normal(name: str, seed: int, shape, std: float = 1.0, mean: float = 0.0) -> np.ndarray:
    """Approximately normal (Irwin-Hall of 4 uniforms, variance-matched): exact fp64 ops only."""
    n = int(np.prod(shape))
    s = uniform01(name, seed, n, 0) + uniform01(name, seed, n, 1) \
        + uniform01(name, seed, n, 2) + uniform01(name, seed, n, 3)
    z = (s - 2.0) * 1.7320508075688772  # var of sum of 4 U(0,1) is 1/3
    return (mean + std * z).astype(np.float32).reshape(shape)


def keep_mask(name: str, seed: int, shape) -> np.ndarray:
    """Bernoulli(0.5) dropout keep-mask as uint8 (1 = keep)."""
    n = int(np.prod(shape))
    return (uniform01(name, seed, n) >= 0.5).astype(np.uint8).reshape(shape)


# --------------------------------------------------------------------------------------
# parameter shapes of the reference nets (cgan/models.py:89-123, :222-253)
# --------------------------------------------------------------------------------------
D_CONVS = [(6, 64), (64, 128), (128, 256), (256, 512)]        # SN conv k4 s2 p1 + bias
D_IDX = [0, 2, 5, 8]                                          # nn.Sequential positions
G_DOWN = [(3, 64), (64, 128), (128, 256), (256, 512)]         # Conv2d weight [Cout,Cin,4,4]
G_UP = [(512, 256), (512, 128), (256, 64), (128, 64)]         # ConvTranspose2d weight [Cin,Cout,4,4]


def discriminator_state(seed: int, spectral_norm: bool = True) -> dict:
    """state_dict-shaped numpy arrays for Discriminator(spectral_norm=True); with spectral_norm=False the same values under the
    plain-conv keys ``model.N.weight`` (no u, v): ``conv_block`` without the hook (cgan/models.py:235-243).

    Conv weights ~N(0,0.02) and zero biases as ``weights_init_normal`` leaves them
    (cgan/models.py:37-48); u,v are unit vectors as ``spectral_norm`` initialises them.
    A small non-zero bias is used for model.0 so bias handling is actually exercised.
    """
    sd = {}
    for (cin, cout), i in zip(D_CONVS, D_IDX):
        sd[f"model.{i}.weight_orig"] = normal(f"D.{i}.w", seed, (cout, cin, 4, 4), 0.02)
        sd[f"model.{i}.bias"] = normal(f"D.{i}.b", seed, (cout,), 0.01)
        u = normal(f"D.{i}.u", seed, (cout,), 1.0).astype(np.float64)
        v = normal(f"D.{i}.v", seed, (cin * 16,), 1.0).astype(np.float64)
        sd[f"model.{i}.weight_u"] = (u / np.sqrt((u * u).sum())).astype(np.float32)
        sd[f"model.{i}.weight_v"] = (v / np.sqrt((v * v).sum())).astype(np.float32)
    sd["model.11.weight"] = normal("D.11.w", seed, (1, 512, 4, 4), 0.02)
    if not spectral_norm:
        sd = {k.replace("weight_orig", "weight"): v for k, v in sd.items() if not k.endswith(("weight_u", "weight_v"))}
    return sd


def generator_state(seed: int) -> dict:
    sd = {}
    for k, (cin, cout) in enumerate(G_DOWN, 1):
        sd[f"down{k}.model.0.weight"] = normal(f"G.down{k}", seed, (cout, cin, 4, 4), 0.02)
    for k, (cin, cout) in enumerate(G_UP, 1):
        key = f"up{k}.model.0.weight" if k < 4 else "up4.0.weight"
        sd[key] = normal(f"G.up{k}", seed, (cin, cout, 4, 4), 0.02)
    # nn.Linear default init range is +-1/sqrt(64)
    sd["fc_delta.1.weight"] = uniform("G.fc.w", seed, (4, 64), -0.125, 0.125)
    sd["fc_delta.1.bias"] = uniform("G.fc.b", seed, (4,), -0.125, 0.125)
    return sd


GS_CONV = [(3, 64), (64, 64), (64, 128), (128, 128), (128, 256), (256, 256), (256, 512), (512, 512)]
GS_CONV_IDX = (0, 3, 7, 10, 14, 17, 21, 24)
GS_FC = [(512, 256), (256, 64), (64, 4)]
GS_FC_IDX = (2, 5, 8)


def simple_generator_state(seed: int) -> dict:
    """state_dict-shaped numpy arrays for GeneratorSimpleRegressor (cgan/models.py:147-216): conv weights N(0, 0.02) as
    weights_init_normal leaves them (models.py:37-48), conv biases and the Linear layers in their default uniform ranges."""
    sd = {}
    for j, ((cin, cout), i) in enumerate(zip(GS_CONV, GS_CONV_IDX)):
        sd[f"features.{i}.weight"] = normal(f"GS.conv{j}", seed, (cout, cin, 3, 3), 0.02)
        bound = 1.0 / np.sqrt(cin * 9)
        sd[f"features.{i}.bias"] = uniform(f"GS.conv{j}.b", seed, (cout,), -bound, bound)
    for j, ((fin, fout), i) in enumerate(zip(GS_FC, GS_FC_IDX)):
        bound = 1.0 / np.sqrt(fin)
        sd[f"regressor.{i}.weight"] = uniform(f"GS.fc{j}.w", seed, (fout, fin), -bound, bound)
        sd[f"regressor.{i}.bias"] = uniform(f"GS.fc{j}.b", seed, (fout,), -bound, bound)
    return sd


def gs_mask_shapes(batch: int):
    """Shapes of the two Dropout(0.5) sites of GeneratorSimpleRegressor.regressor (cgan/models.py:205,208)."""
    return [(batch, 256), (batch, 64)]


def g_mask_shapes(batch: int, size: int):
    """Shapes (NCHW) of the three Dropout(0.5) sites in GeneratorUNet (cgan/models.py:106,109,110)."""
    return [(batch, 512, size // 16, size // 16),
            (batch, 256, size // 8, size // 8),
            (batch, 128, size // 4, size // 4)]


def step_inputs(seed: int, batch: int, size: int, n_critic: int = 2, tag: str = "", generator_type: str = "unet") -> dict:
    """One iteration's worth of inputs (numpy, NCHW fp32) for the cGAN step.

    ``refined``: one tensor per critic step plus one for the G step (stand-ins for the host
    re-crop stage cgan/cgan_train_enhanced.py:37-137, which has no synthetic equivalent).
    ``alpha``: one (B,1,1,1) per critic step (cgan/losses.py:199).
    ``masks``: (n_critic + 1) x 3 keep-masks, consumed in the order G is called.
    """
    t = f"{tag}/B{batch}S{size}"
    out = {
        "pred": uniform(f"{t}/pred", seed, (batch, 3, size, size), -1.0, 1.0),
        "gt": uniform(f"{t}/gt", seed, (batch, 3, size, size), -1.0, 1.0),
        "pred_box": np.concatenate([uniform(f"{t}/box.c", seed, (batch, 2), 0.3, 0.7),
                                    uniform(f"{t}/box.s", seed, (batch, 2), 0.1, 0.5)], axis=1),
        "delta_true": np.clip(normal(f"{t}/dtrue", seed, (batch, 4), 0.1), -2.3025851, 2.3025851),
        "refined": [uniform(f"{t}/refined{k}", seed, (batch, 3, size, size), -1.0, 1.0)
                    for k in range(n_critic + 1)],
        "alpha": [uniform(f"{t}/alpha{k}", seed, (batch, 1, 1, 1)) for k in range(n_critic)],
        "masks": [[keep_mask(f"{t}/mask{k}.{j}", seed, shp)
                   for j, shp in enumerate(gs_mask_shapes(batch) if generator_type == "simple" else g_mask_shapes(batch, size))]
                  for k in range(n_critic + 1)],
    }
    return out


def sample_indices(numel: int, count: int = 256) -> np.ndarray:
    """Deterministic strided sample positions used by the fixtures to pin big tensors."""
    if numel <= count:
        return np.arange(numel, dtype=np.int64)
    step = numel / float(count)
    return np.minimum((np.arange(count) * step).astype(np.int64), numel - 1)
