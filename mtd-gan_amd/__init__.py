"""mtd_gan_amd -- MI355X-native (gfx950) implementation of the MTD-GAN training hot path.

Package directory is `mtd-gan_amd/`; import it as `mtd_gan_amd` (see the shim at the repo root).
Layout: csrc/ (HIP kernels + C ABI, built into libmtdgan_hip.so), kernels.py (launch helpers),
generator_path.py / discriminator_path.py (explicit forward/backward kernel schedules) and the mirror
of the reference's module surface: arch/Ours/networks.py, losses.py, module/weight_methods.py, engine.py.
"""
__version__ = "0.1.0"
