"""The C-ABI shared library loads (no GPU needed) and exports every symbol include/mixq_hip.h declares; the ctypes
signature table mirrors the header; argument validation is exercised where it needs no device."""
import ctypes as C
import os
import re
import subprocess

import pytest

from mixq_amd import _capi


def test_library_is_built_in_tree():
    assert os.path.exists(_capi.LIB_PATH), "run __graft_entry__.build() first"
    assert os.path.dirname(_capi.LIB_PATH).endswith("mixq_amd")


def test_every_header_symbol_is_exported_and_bound():
    lib = _capi.load()
    declared = _capi.header_symbols()
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/mixq_hip.h but not exported"
    assert set(declared) == set(_capi.SIGNATURES) - set(_capi.TUNING_ONLY), "ctypes table and header disagree"
    assert set(_capi.header_symbols(tuning=True)) == set(_capi.SIGNATURES)       # ... and with the tuning library's section


def test_header_arity_matches_ctypes_table():
    txt = open(_capi.HEADER_PATH).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    for m in re.finditer(r"\b(?:int|long long)\s+(mixq_\w+)\s*\(([^)]*)\)\s*;", txt):
        name, args = m.group(1), m.group(2).strip()
        n = 0 if args in ("", "void") else len(args.split(","))
        assert n == len(_capi.SIGNATURES[name]), f"{name}: header has {n} parameters, ctypes table {len(_capi.SIGNATURES[name])}"


def test_linear_args_block_matches_the_header():
    """struct mixq_linear_args: the ctypes mirror has the header's fields in the header's order with matching C types."""
    txt = open(_capi.HEADER_PATH).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    body = re.search(r"typedef struct mixq_linear_args \{(.*?)\} mixq_linear_args;", txt, flags=re.S).group(1)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        kind = "ptr" if "*" in decl else ("float" if decl.startswith("float") else "int")
        names = [v.strip().lstrip("*").strip() for v in re.sub(r"^(const\s+)?\w+\s*\**", "", decl, count=1).split(",")]
        fields += [(nm, kind) for nm in names]
    want = {"ptr": C.c_void_p, "int": C.c_int, "float": C.c_float}
    assert [(nm, want[k]) for nm, k in fields] == list(_capi.LinearArgs._fields_)
    assert C.sizeof(_capi.LinearArgs) >= 8 * 14 + 4 * 13


def test_code_object_is_gfx950():
    """The fat binary carries gfx950 code objects and nothing for another GPU (read from the bundle's own entry ids; running
    llvm-objdump --offloading would unpack the bundle next to the library)."""
    blob = open(_capi.LIB_PATH, "rb").read()
    targets = set(re.findall(rb"hipv4-amdgcn-amd-amdhsa--(gfx[0-9a-z]+)", blob))
    assert targets == {b"gfx950"}, targets


def test_version_and_config_table():
    lib = _capi.load()
    assert lib.mixq_version() >= 1000
    names = _capi.gemm_config_names()
    assert len(names) == lib.mixq_gemm_num_configs() and len(set(names)) == len(names)
    assert lib.mixq_gemm_set_config(len(names)) == _capi.MIXQ_EINVAL
    assert lib.mixq_gemm_set_config(-1) == 0
    c = lib.mixq_gemm_pick_config(512, 11008, 4096, 8)
    assert 0 <= c < len(names)
    assert lib.mixq_gemm_pick_config(0, 1, 1, 8) == _capi.MIXQ_EINVAL


def test_argument_validation_without_device():
    """Bad arguments are rejected before anything touches the GPU."""
    lib = _capi.load()
    one = C.c_void_p(16)   # a non-null dummy pointer; never dereferenced on these paths
    assert lib.mixq_find_row_scale(None, one, one, 4, 64, 64, 8, 0, None) == _capi.MIXQ_EINVAL
    assert lib.mixq_find_row_scale(one, one, one, 4, 64, 64, 5, 0, None) == _capi.MIXQ_EINVAL      # bit
    assert lib.mixq_find_row_scale(one, one, one, 4, 60, 64, 8, 0, None) == _capi.MIXQ_ESHAPE      # K % 8
    assert lib.mixq_find_row_scale(one, one, one, 4, 72, 72, 8, 1, None) == _capi.MIXQ_ESHAPE      # packed needs K % 64
    assert lib.mixq_find_row_scale(one, one, one, 0, 64, 64, 8, 0, None) == 0                      # empty input is fine
    assert lib.mixq_gemm_i8_fused(one, one, one, one, None, 0, None, 0, 0, None, None, 0, None, one, 64, 4, 64, 100, 0, 0,
                                  None) == _capi.MIXQ_ESHAPE                                        # K % 64
    assert lib.mixq_gemm_i8_fused(one, one, one, one, None, 0, None, 0, 0, None, None, 0, None, one, 64, 4, 64, 128, 7, 0,
                                  None) == _capi.MIXQ_EINVAL                                        # act
    assert lib.mixq_gemm_i8_fused(one, one, one, one, None, 0, None, 0, 0, None, None, 0, None, one, 64, 0, 64, 128, 0, 0,
                                  None) == 0                                                        # M = 0
    assert lib.mixq_pack_p16x64(one, one, 4, 100, None) == _capi.MIXQ_ESHAPE
    assert lib.mixq_extract_outliers_zero(one, None, 3, one, 4, 64, 64, 3, None) == _capi.MIXQ_EINVAL
    # weight-only W8A16
    assert lib.mixq_pack_w8a16(one, one, 100, 64, None) == _capi.MIXQ_ESHAPE                        # K % 64
    assert lib.mixq_pack_w8a16(None, one, 64, 64, None) == _capi.MIXQ_EINVAL
    assert lib.mixq_gemm_w8a16(one, 128, one, one, None, one, 64, 4, 64, 100, None) == _capi.MIXQ_ESHAPE      # K % 64
    assert lib.mixq_gemm_w8a16(one, 128, one, one, None, one, 64, 4, 62, 128, None) == _capi.MIXQ_ESHAPE      # N % 4
    assert lib.mixq_gemm_w8a16(one, 100, one, one, None, one, 64, 4, 64, 128, None) == _capi.MIXQ_ESHAPE      # ldx < K
    assert lib.mixq_gemm_w8a16(C.c_void_p(8), 128, one, one, None, one, 64, 4, 64, 128, None) == _capi.MIXQ_EINVAL  # x alignment
    assert lib.mixq_gemm_w8a16(one, 128, one, one, None, one, 64, 0, 64, 128, None) == 0                      # M = 0
    assert lib.mixq_gemm_w8a16_set_config(99) == _capi.MIXQ_EINVAL and lib.mixq_gemm_w8a16_set_config(-1) == 0
    # the one-call forward checks its block before it launches anything
    assert lib.mixq_linear_forward(None, None) == _capi.MIXQ_EINVAL
    blk = _capi.LinearArgs()
    blk.bit = 3
    assert lib.mixq_linear_forward(C.byref(blk), None) == _capi.MIXQ_EINVAL


def test_product_library_ships_no_tuning_variants():
    """The product .so exports no configuration whose output is wrong by design (the ablation forms), no trace stamps, no K rotation
    and no quantise probes: those live in the -DMIXQ_TUNING build only (engineering #8)."""
    assert os.path.basename(_capi.LIB_PATH) == "libmixq_hip.so"
    lib = _capi.load()
    assert all("abl" not in nm for nm in _capi.gemm_config_names()), _capi.gemm_config_names()
    assert all("abl" not in nm for nm in _capi.w8a16_config_names())
    for name in _capi.TUNING_ONLY:                                          # trace stamps / tile-order knob: not even an entry point
        assert not hasattr(lib, name), name
    assert lib.mixq_quant_set_config(100) == _capi.MIXQ_EINVAL and lib.mixq_quant_set_config(-1) == 0
    blob = open(_capi.LIB_PATH, "rb").read()
    assert b"abl3_mfma" not in blob and b"abl1_noW" not in blob


def test_forced_configuration_round_trips():
    lib = _capi.load()
    names = _capi.gemm_config_names()
    assert lib.mixq_gemm_set_config(len(names) - 1) == 0 and lib.mixq_gemm_set_config(-1) == 0
    assert lib.mixq_quant_set_config(9) == 0 and lib.mixq_quant_set_config(10) == _capi.MIXQ_EINVAL and lib.mixq_quant_set_config(-1) == 0


def test_forced_rebuild_of_one_object(tmp_path):
    """`make -B` on one object (the forced-rebuild path of __graft_entry__.build(force=True), on the smallest source): hipcc cross-
    compiles gfx950 from the tracked sources without a GPU, and the object carries the entry point."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    obj = os.path.join(root, "build", "obj", "forward.o")
    r = subprocess.run(["make", "-C", os.path.join(root, "mixq_amd", "csrc"), "-B", "../../build/obj/forward.o"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-1500:]
    assert "hipcc" in r.stdout and os.path.exists(obj) and b"mixq_linear_forward" in open(obj, "rb").read()


def test_missing_library_is_loud(monkeypatch, tmp_path):
    monkeypatch.setattr(_capi, "_lib", None)
    monkeypatch.setattr(_capi, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_capi.MixqBuildError):
        _capi.load()
