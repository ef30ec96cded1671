"""Per-model switches of the MI355X operator, captured at construction.

Rounds 1-4 kept these as seven process-global module variables (`linear.PACK_FMT`, `fused.JOINT_GATE_UP`, ...): two models in one process
changed each other's kernels and cache keys.  Now a `MixqConfig` lives on the model's `MixLibCache` (`cache.config`; the reference builds
ONE cache per model and hands it to every layer, mixquant/Cache.py:5, models/llama.py) and every layer / fused module keeps a reference to
the config of the cache it was built with - a layer built without a cache gets its own.  The reference has no counterpart: its only
switches are constructor arguments of the same objects."""
from __future__ import annotations

from dataclasses import dataclass, replace

from ._capi import FMT_F16X64, FMT_F6X128


@dataclass
class MixqConfig:
    # Packed layout 8-bit layers keep their weights in (and ask their activations in): MIXQ_FMT_F16X64 feeds the weights-in-registers GEMM
    # of gemm_wreg.hip, MIXQ_FMT_P16X64 the LDS-staged one of gemm.hip (include/mixq_hip.h).
    pack_fmt: int = FMT_F16X64
    # ... and 4-bit layers: MIXQ_FMT_F6X128 - both operands as FP6 E3M2 codes, the W4A4 GEMM on the FP6 matrix pipe (gfx950 has no int4
    # MFMA; every integer of [-8, 8] is an E3M2 value and the fp32 accumulator is exact, so the result is the int4 contraction bit for bit
    # at 1.6x the int8 MFMA rate).  FMT_P16X64 selects the int8-expansion kernel instead (nibbles, two thirds of the FP6 image's bytes).
    pack_fmt4: int = FMT_F6X128
    # ONE resident weight image per 4-bit layer: the FP6 one (0.75 byte per weight).  A decode-heavy deployment of NARROW layers may opt
    # into a SECOND, nibble image for batches of at most this many rows (32 is the useful value; profiles/r04_w4a4_small_batch.txt).  0 = off.
    small_batch_m4: int = 0
    # After a layer's outlier search has frozen, keep ONLY the packed weight image in HBM (the plain [N,K] `q_weight` is re-created on
    # demand for state_dict / attribute reads).  False keeps both copies (2x the reference's weight memory).
    compact_weights: bool = True
    # Frozen layers run their forward (extract + quantise, GEMM) through ONE C-ABI call on a kept argument block (mixq_linear_forward).
    # False keeps the two-call route (same kernels, same results; the parity tests compare the two).
    one_call_forward: bool = True
    # gate_proj's GEMM also emits down_proj's per-row |x| maxima (mixq_gemm_i8_fused_amax); False = the two-pass quantiser
    fuse_down_amax: bool = True
    # gate_proj + up_proj of MixLlamaMLP as one launch over an interleaved weight image once both predictions are frozen; False = up_proj's
    # launch, then gate_proj's with the SiLU-and-multiply epilogue
    joint_gate_up: bool = True
    # the fused norm hands the next (frozen) layer's kept outlier map to its quantise step (mixq_rmsnorm_quant_fused_masked); False = the
    # column mask is built inside the kernel on every launch
    norm_kept_map: bool = True

    def copy(self, **changes) -> "MixqConfig":
        return replace(self, **changes)

    def key(self):
        """What of the configuration is part of a kept plan's / joint image's identity."""
        return (self.pack_fmt, self.pack_fmt4, self.small_batch_m4)
