"""clibd_amd — MI355X (gfx950) implementation of CLIBD's contrastive training step.

Drop-in for the reference's `bioscanclip.model` encoder/loss API (see `clibd_amd.model`), running on
hand-written HIP kernels bound through a C ABI (`include/clibd_hip.h`, `libclibd_hip.so`).
"""
__version__ = "0.1.0"
