"""secp256k1_zkp_amd -- MI355X (gfx950) batch-verification engine for the secp256k1-zkp MSM / double-mult hot path.

Python here is plumbing only (ctypes over the C ABI in ``include/secp256k1_zkp_amd.h``, torch for HBM buffers and
``torch.distributed``); all arithmetic runs in the hand-written HIP kernels under ``csrc/``.
"""
from .api import Engine, Group, S2KError  # noqa: F401
