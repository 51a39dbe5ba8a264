"""gcssl-mi355x: MI355X-native (gfx950) implementation of the WGAN-GP cGAN training step of
1213ray/GAN-Calibrated-Semi-Supervised-Learning (reference hot path: cgan/cgan_train_enhanced.py:304-369).

Sub-modules: ``_lib`` (C-ABI loader), ``ops`` (launch wrappers), ``engine`` (explicit-schedule step engine),
``models`` / ``losses`` (the reference's Python surface), ``dist`` (data parallel), ``synth`` (synthetic inputs).
Import is cheap and GPU-free; the HIP library is loaded on first use and there is no CPU fallback.
"""
__version__ = "0.1.0"
