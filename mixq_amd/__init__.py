"""mixq_amd — MI355X-native (gfx950 / CDNA4) implementation of MixQ's mixed-precision quantized Linear hot path.

Public surface (mirrors the reference operator API):
    MixLinear_GEMM (alias MixQLinear), MixLibCache, MLPCache, pack_to_i4, two_compl, unpack_int8_to_int4,
    `mixq_amd.mixlib`, a drop-in for the reference's native `mixlib` module backed by libmixq_hip.so, and
    `mixq_amd.checkpoint`: the reference's checkpoint layout, layer policy, loader and QKV fusion.
"""
from .config import MixqConfig
from .cache import MixLibCache, MLPCache
from .linear import MixLinear_GEMM, MixQLinear, pack_to_i4, two_compl, unpack_int8_to_int4
from . import mixlib
from .mixlib import ShimConfig
from .fused import FasterTransformerRMSNorm, MixFalconMLP, MixGPTJMLP, MixLlamaMLP
from . import checkpoint

__all__ = ["MixqConfig", "MixLinear_GEMM", "MixQLinear", "MixLibCache", "MLPCache", "pack_to_i4", "two_compl", "unpack_int8_to_int4",
           "mixlib", "ShimConfig", "FasterTransformerRMSNorm", "MixLlamaMLP", "MixFalconMLP", "MixGPTJMLP", "checkpoint"]
__version__ = "0.1.0"
