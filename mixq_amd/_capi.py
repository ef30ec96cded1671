"""ctypes binding of libmixq_hip.so — the C-ABI boundary declared in include/mixq_hip.h.

There is NO fallback: if the HIP library is missing or fails to load, importing anything that computes raises
`MixqBuildError`.  (The oracle under oracle/ is test infrastructure and is never imported from here.)
"""
from __future__ import annotations

import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmixq_hip.so")
# tools/ scripts that need the ablation kernels / trace stamps set MIXQ_TUNING_LIB=1 to load the -DMIXQ_TUNING build instead
# (`make -C mixq_amd/csrc tuning`); the package itself never asks for it
if os.environ.get("MIXQ_TUNING_LIB") == "1":
    LIB_PATH = os.path.join(_HERE, "libmixq_hip_tuning.so")
if os.environ.get("MIXQ_LIB_FILE"):                     # development aid: time another BUILD of the library (before / after A/B runs)
    LIB_PATH = os.path.join(_HERE, os.path.basename(os.environ["MIXQ_LIB_FILE"]))
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "mixq_hip.h")

MIXQ_OK = 0
MIXQ_EINVAL = -1
MIXQ_ESHAPE = -2
MIXQ_ENODEV = -3
ACT_NONE = 0
ACT_SILU = 1
ACT_SILU_MUL = 2
ACT_SILU_PAIR = 3                    # gate_proj + up_proj as one GEMM over interleaved rows: y has N / 2 columns (include/mixq_hip.h)
FMT_PLAIN = 0
FMT_P16X64 = 1
FMT_F16X64 = 2
FMT_F6X128 = 3          # int4 as FP6 codes, fragment order: the weights of the W4A4 GEMM on the FP6 matrix pipe
FMT_R6X128 = 4          # ... and its activations: the same fragments, a row's 96 bytes of a block kept together
X_PACKED = 1
W_PACKED = 2
W_F16X64 = 8
XW_F6X128 = 16


class MixqBuildError(RuntimeError):
    pass


class MixqError(RuntimeError):
    def __init__(self, fn: str, code: int):
        names = {MIXQ_EINVAL: "MIXQ_EINVAL (bad argument)", MIXQ_ESHAPE: "MIXQ_ESHAPE (unsupported shape)",
                 MIXQ_ENODEV: "MIXQ_ENODEV (no gfx950 device)"}
        super().__init__(f"{fn} failed: {names.get(code, f'hipError_t {code}')}")
        self.code = code


_P = C.c_void_p
_I = C.c_int
_F = C.c_float

# name -> argtypes; mirrors include/mixq_hip.h one-to-one (tests/test_capi_symbols.py parses the header and checks)
SIGNATURES = {
    "mixq_version": [],
    "mixq_device_info": [C.c_char_p, _I],
    "mixq_find_row_scale": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "mixq_extract_outliers_zero": [_P, _P, _I, _P, _I, _I, _I, _I, _P],
    "mixq_quant_fused": [_P, _P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _P],
    "mixq_quant_fused_masked": [_P, _P, _I, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _P],
    "mixq_kept_map_words": [_I],
    "mixq_detect_outlier_cols": [_P, _F, _P, _P, _P, _I, _I, _I, _P],
    "mixq_dequant_weight_cols": [_P, _P, _P, _I, _P, _I, _I, _I, _I, _P],
    "mixq_gemm_i8_fused": [_P, _P, _P, _P, _P, _I, _P, _I, _I, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "mixq_gemm_i4_fused": [_P, _P, _P, _P, _P, _I, _P, _I, _I, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "mixq_gemm_i8": [_P, _P, _P, _I, _I, _I, _I, _P],
    "mixq_dequant": [_P, _I, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _P],
    "mixq_gemm_set_config": [_I],
    "mixq_gemm_set_trace": [_P],
    "mixq_gemm_set_krot": [_I],
    "mixq_selftest_quant_exact": [_P, _I, _P],
    "mixq_gemm_num_configs": [],
    "mixq_pack_p16x64": [_P, _P, _I, _I, _P],
    "mixq_pack_operand": [_P, _P, _I, _I, _I, _P],
    "mixq_unpack_operand": [_P, _P, _I, _I, _I, _P],
    "mixq_quant_set_config": [_I],
    "mixq_gemm_config_name": [_I, C.c_char_p, _I],
    "mixq_gemm_pick_config": [_I, _I, _I, _I],
    "mixq_gemm_pick_config_fmt": [_I, _I, _I, _I, _I],
    "mixq_gemm_pick_split": [_I, _I, _I, _I, _I, _P, _P],
    "mixq_rmsnorm": [_P, _P, _P, _I, _I, _I, _I, _F, _P],
    "mixq_rmsnorm_quant_fused": [_P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _F, _I, _P],
    "mixq_rmsnorm_quant_fused_masked": [_P, _P, _P, _P, _I, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _F, _I, _P],
    "mixq_pack_w8a16": [_P, _P, _I, _I, _P],
    "mixq_gemm_w8a16": [_P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "mixq_gemm_w8a16_set_config": [_I],
    "mixq_gemm_w8a16_num_configs": [],
    "mixq_gemm_w8a16_config_name": [_I, C.c_char_p, _I],
    "mixq_gemm_workspace_bytes": [],
    "mixq_gemm_set_workspace": [_P, C.c_longlong],
    "mixq_gemm_hint_next_weights": [_P, C.c_longlong],
    "mixq_gemm_set_fuse_probe": [_P, _P, _P, _I],
    "mixq_linear_forward": [_P, _P],
    "mixq_gemm_i8_fused_amax": [_P, _P, _P, _P, _P, _I, _P, _I, _I, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P],
    "mixq_gemm_amax_supported": [_I, _I, _I, _I],
    "mixq_quant_known_amax": [_P, _P, _I, _P, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _P],
}
RESTYPES = {"mixq_gemm_workspace_bytes": C.c_longlong}


class LinearArgs(C.Structure):
    """struct mixq_linear_args of include/mixq_hip.h, field for field (tests/test_capi_symbols.py compares the two)."""
    _fields_ = [("x", _P), ("ldx", _I),
                ("ind", _P), ("n_cap", _I), ("n_dev", _P),
                ("x_scale", _P), ("q_x", _P), ("x_out", _P), ("ldxo", _I), ("flag", _P),
                ("q_w", _P), ("scale_col", _P), ("w_out", _P), ("ldwo", _I),
                ("addend", _P), ("lda", _I), ("bias", _P),
                ("y", _P), ("ldy", _I),
                ("M", _I), ("N", _I), ("K", _I), ("bit", _I), ("sigma", _F), ("act", _I), ("qfmt", _I), ("wfmt", _I),
                ("row_amax", _P), ("col_mask", _P), ("col_mask_words", _I)]

_lib = None


TUNING_ONLY = ("mixq_gemm_set_trace", "mixq_gemm_set_krot", "mixq_gemm_hint_next_weights", "mixq_gemm_set_fuse_probe")     # exported by libmixq_hip_tuning.so only (include/mixq_hip.h: #ifdef MIXQ_TUNING)


def header_symbols(path: str = HEADER_PATH, tuning: bool = False):
    """Function names declared in include/mixq_hip.h (used by the symbol-export test); the `#ifdef MIXQ_TUNING` section counts only for the
    tuning library."""
    txt = open(path).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    if not tuning:
        txt = re.sub(r"#ifdef MIXQ_TUNING.*?#endif", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|long long)\s+(mixq_\w+)\s*\(", txt)))


def load():
    """Load the shared library once; raise MixqBuildError loudly if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MixqBuildError(
            f"{LIB_PATH} not found - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  mixq_amd has no CPU fallback.")
    try:
        import torch  # noqa: F401  (makes torch's libamdhip64.so.7 the HIP runtime this library binds to)
    except Exception:  # pragma: no cover
        pass
    try:
        lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    except OSError as e:
        raise MixqBuildError(f"cannot load {LIB_PATH}: {e}") from e
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            if name in TUNING_ONLY:
                continue                                   # the product library: no trace stamps, no tile-order knob
            raise MixqBuildError(f"{LIB_PATH} does not export {name}; rebuild it") from e
        fn.argtypes = argtypes
        fn.restype = RESTYPES.get(name, _I)
    _lib = lib
    return lib


def call(name: str, *args) -> None:
    """Call an entry point and raise MixqError on a non-zero status."""
    rc = getattr(load(), name)(*args)
    if rc != 0:
        raise MixqError(name, rc)


def device_info() -> str:
    buf = C.create_string_buffer(256)
    rc = load().mixq_device_info(buf, 256)
    if rc != 0:
        raise MixqError("mixq_device_info", rc)
    return buf.value.decode()


_workspaces = {}


def ensure_workspace(device=None):
    """Allocate (once per process and device) and register the zero-filled stream-K workspace on that GPU."""
    import torch
    dev = torch.device(device or "cuda")
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if idx not in _workspaces:
        lib = load()
        nbytes = int(lib.mixq_gemm_workspace_bytes())
        with torch.cuda.device(idx):
            ws = torch.zeros(nbytes, dtype=torch.uint8, device=f"cuda:{idx}")
            rc = lib.mixq_gemm_set_workspace(ws.data_ptr(), nbytes)      # registered for the device that is current here
        if rc != 0:
            raise MixqError("mixq_gemm_set_workspace", rc)
        _workspaces[idx] = ws
    return _workspaces[idx]


def w8a16_config_names():
    lib = load()
    out = []
    for i in range(lib.mixq_gemm_w8a16_num_configs()):
        buf = C.create_string_buffer(64)
        lib.mixq_gemm_w8a16_config_name(i, buf, 64)
        out.append(buf.value.decode())
    return out


def gemm_split_plan(M, N, K, bit=8, fmt=2):
    """(width of the first launch's column range, name of the second launch's tiling) when the automatic choice splits (M, N, K) along N
    (include/mixq_hip.h: mixq_gemm_pick_split), else None."""
    lib = load()
    n1, c2 = C.c_int(0), C.c_int(-1)
    rc = lib.mixq_gemm_pick_split(M, N, K, bit, fmt, C.byref(n1), C.byref(c2))
    if rc < 0:
        raise RuntimeError(f"mixq_gemm_pick_split: status {rc}")
    return (n1.value, gemm_config_names()[c2.value]) if rc == 1 else None


def gemm_config_names():
    lib = load()
    out = []
    for i in range(lib.mixq_gemm_num_configs()):
        buf = C.create_string_buffer(64)
        lib.mixq_gemm_config_name(i, buf, 64)
        out.append(buf.value.decode())
    return out
