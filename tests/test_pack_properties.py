"""Property tests (hypothesis) for the bit-exact integer parts of the path: nibble packing, the P16x64 tile-major
layout, and the quantisation invariants of the oracle."""
import numpy as np
import torch
from hypothesis import given, settings, strategies as st

from mixq_amd import pack_to_i4
from oracle import oracle as O


def p16x64_reference(q):
    """Straight restatement of the P16x64 layout of include/mixq_hip.h: [KB/64][rows16/16] blocks of 16 rows x 64 B,
    the four 16-byte chunks of row r stored at position c ^ (-(r>>2) & 3); rows >= R zero."""
    R, KB = q.shape
    rows16 = (R + 15) // 16 * 16
    out = np.zeros(rows16 * KB, dtype=np.uint8)
    src = q.view(np.uint8)
    for row in range(R):
        for kb in range(KB // 64):
            base = (kb * (rows16 // 16) + row // 16) * 1024 + (row % 16) * 64
            for c in range(4):
                pc = c ^ ((-((row % 16) >> 2)) & 3)
                out[base + pc * 16: base + pc * 16 + 16] = src[row, kb * 64 + c * 16: kb * 64 + c * 16 + 16]
    return out


def p16x64_unpack(buf, R, KB):
    rows16 = (R + 15) // 16 * 16
    out = np.zeros((R, KB), dtype=np.uint8)
    for row in range(R):
        for kb in range(KB // 64):
            base = (kb * (rows16 // 16) + row // 16) * 1024 + (row % 16) * 64
            for c in range(4):
                pc = c ^ ((-((row % 16) >> 2)) & 3)
                out[row, kb * 64 + c * 16: kb * 64 + c * 16 + 16] = buf[base + pc * 16: base + pc * 16 + 16]
    return out


def f16x64_reference(q):
    """MIXQ_FMT_F16X64 of include/mixq_hip.h: the same grid of 1 KiB blocks, inside a block byte c*256 + r*16 + b for row r,
    16-byte k-chunk c: lane l of a wave reading 16 bytes at block + 16 l gets row l & 15, chunk l >> 4."""
    R, KB = q.shape
    rows16 = (R + 15) // 16 * 16
    out = np.zeros(rows16 * KB, dtype=np.uint8)
    src = q.view(np.uint8)
    for row in range(R):
        for kb in range(KB // 64):
            base = (kb * (rows16 // 16) + row // 16) * 1024
            for c in range(4):
                o = base + c * 256 + (row % 16) * 16
                out[o: o + 16] = src[row, kb * 64 + c * 16: kb * 64 + c * 16 + 16]
    return out


def f16x64_unpack(buf, R, KB):
    rows16 = (R + 15) // 16 * 16
    out = np.zeros((R, KB), dtype=np.uint8)
    for row in range(R):
        for kb in range(KB // 64):
            base = (kb * (rows16 // 16) + row // 16) * 1024
            for c in range(4):
                o = base + c * 256 + (row % 16) * 16
                out[row, kb * 64 + c * 16: kb * 64 + c * 16 + 16] = buf[o: o + 16]
    return out


F6_MAG = (0x00, 0x0C, 0x10, 0x12, 0x14, 0x15, 0x16, 0x17, 0x18)      # E3M2 codes of 0..8 (sign bit 0x20): 1 2 (1.5 x 2) 4 (1.25 x 4) (1.5 x 4) (1.75 x 4) 8


def f6_value(code):
    """The value an FP6 E3M2 code denotes (OCP MX: 1 sign, 3 exponent bits of bias 3, 2 mantissa bits; no infinities, no NaNs)."""
    s, e, m = code >> 5, (code >> 2) & 7, code & 3
    v = (m / 4.0) * 2.0 ** -2 if e == 0 else (1 + m / 4.0) * 2.0 ** (e - 3)
    return -v if s else v


def f6x128_reference(v):
    """MIXQ_FMT_F6X128 of include/mixq_hip.h restated byte by byte: v int8 [R, K] with values of [-8, 7] -> the image, [K/128][rows16/16]
    blocks of 1536 bytes; lane l = 16 (k % 128 // 32) + row % 16 owns elements 32 (l >> 4) .. + 31 as a little-endian stream of 6-bit
    codes, its first 16 bytes at block + 16 l, the last 8 at block + 1024 + 8 l; rows >= R hold the code of 0."""
    R, K = v.shape
    rows16 = (R + 15) // 16 * 16
    out = np.zeros((K // 128) * (rows16 // 16) * 1536, dtype=np.uint8)
    for row in range(R):
        for kb in range(K // 128):
            base = (kb * (rows16 // 16) + row // 16) * 1536
            for g in range(4):
                lane = g * 16 + row % 16
                bits = 0
                for e in range(32):
                    x = int(v[row, kb * 128 + g * 32 + e])
                    bits |= ((0x20 if x < 0 else 0) | F6_MAG[abs(x)]) << (6 * e)
                b = np.frombuffer(bits.to_bytes(24, "little"), dtype=np.uint8)
                out[base + lane * 16: base + lane * 16 + 16] = b[:16]
                out[base + 1024 + lane * 8: base + 1024 + lane * 8 + 8] = b[16:]
    return out


def r6x128_reference(v):
    """MIXQ_FMT_R6X128 (the activation side): the same blocks and 24-byte fragments, row-major inside a block: row r's 96 bytes at 96 r -
    its four 16-byte pieces (16 g), then its four 8-byte pieces (64 + 8 g)."""
    R, K = v.shape
    rows16 = (R + 15) // 16 * 16
    frag = f6x128_reference(v).reshape(K // 128, rows16 // 16, 1536)
    out = np.zeros_like(frag)
    for r in range(16):
        for g in range(4):
            lane = g * 16 + r
            out[:, :, r * 96 + g * 16: r * 96 + g * 16 + 16] = frag[:, :, lane * 16: lane * 16 + 16]
            out[:, :, r * 96 + 64 + g * 8: r * 96 + 64 + g * 8 + 8] = frag[:, :, 1024 + lane * 8: 1024 + lane * 8 + 8]
    return out.reshape(-1)


def r6x128_to_fragment_order(buf, R, K):
    """An R6X128 image re-ordered into F6X128 (for f6x128_unpack)."""
    rows16 = (R + 15) // 16 * 16
    src = np.asarray(buf, dtype=np.uint8).reshape(K // 128, rows16 // 16, 1536)
    out = np.zeros_like(src)
    for r in range(16):
        for g in range(4):
            lane = g * 16 + r
            out[:, :, lane * 16: lane * 16 + 16] = src[:, :, r * 96 + g * 16: r * 96 + g * 16 + 16]
            out[:, :, 1024 + lane * 8: 1024 + lane * 8 + 8] = src[:, :, r * 96 + 64 + g * 8: r * 96 + 64 + g * 8 + 8]
    return out.reshape(-1)


def f6x128_unpack(buf, R, K):
    """Inverse of f6x128_reference through the VALUES the codes denote (so a code that is no integer would show)."""
    rows16 = (R + 15) // 16 * 16
    out = np.zeros((R, K), dtype=np.int8)
    for row in range(R):
        for kb in range(K // 128):
            base = (kb * (rows16 // 16) + row // 16) * 1536
            for g in range(4):
                lane = g * 16 + row % 16
                b = bytes(buf[base + lane * 16: base + lane * 16 + 16]) + bytes(buf[base + 1024 + lane * 8: base + 1024 + lane * 8 + 8])
                bits = int.from_bytes(b, "little")
                for e in range(32):
                    val = f6_value((bits >> (6 * e)) & 63)
                    assert val == int(val)
                    out[row, kb * 128 + g * 32 + e] = int(val)
    return out


def test_every_int4_value_is_an_e3m2_value_and_products_stay_exact():
    """The premise of the FP6 carrier: [-8, 8] embeds exactly into E3M2 - the reference's 4-bit weights are
    clamp(round(w / scale), -8, 7), linear.py:139 - while E2M3 stops at 7.5, and a K-long sum of products stays below 2^24 (exact in the
    fp32 accumulator) for every K the models use."""
    vals = {f6_value(c) for c in range(64)}
    for x in range(-8, 9):
        assert float(x) in vals and f6_value((0x20 if x < 0 else 0) | F6_MAG[abs(x)]) == x
    assert max(vals) == 28.0
    e2m3 = {(m / 8.0 if e == 0 else (1 + m / 8.0) * 2.0 ** (e - 1)) for e in range(4) for m in range(8)}
    assert 8.0 not in e2m3 and max(e2m3) == 7.5
    assert 8 * 7 * 28672 < 2 ** 24 and 8 * 7 * 262144 < 2 ** 24          # |weight| <= 8, |activation| <= 7 (symmetric, qmax = 7)


@settings(max_examples=15, deadline=None)
@given(st.integers(1, 40), st.integers(1, 3), st.integers(0, 2 ** 31 - 1))
def test_f6x128_layout_is_a_bijection_and_host_unpack_inverts_it(rows, kblocks, seed):
    import torch
    from mixq_amd.linear import _unpack_host
    rng = np.random.default_rng(seed)
    v = rng.integers(-8, 8, (rows, 128 * kblocks), dtype=np.int8)
    buf = f6x128_reference(v)
    assert np.array_equal(f6x128_unpack(buf, rows, 128 * kblocks), v)
    rows16 = (rows + 15) // 16 * 16
    back = _unpack_host(torch.from_numpy(buf.reshape(rows16, 96 * kblocks)), rows, 3).numpy()
    assert np.array_equal(back, O.pack_i4(v))
    assert np.array_equal(r6x128_to_fragment_order(r6x128_reference(v), rows, 128 * kblocks), buf)
    # The GEMM's LDS image of an R6X128 block: the block's bytes with the two 16-byte units of every aligned 32-byte pair swapped in rows
    # 8 .. 15 (a DMA lane's source address is free).  Lane (r = l & 15, g = l >> 4) reads its 16-byte piece with ds_read_b128 at
    # 96 r + 16 (g ^ (r >> 3)) and its 8-byte piece with ds_read_b64 at 96 r + 64 + 8 (g ^ 2 (r >> 3)): no two lanes of a service group
    # (16 lanes of a b128 read, 32 of a b64 read: 256 bytes = all 64 banks once) share a bank (MI355X_MICROARCH.md, LDS)
    a_of = lambda l: 96 * (l & 15) + 16 * ((l >> 4) ^ ((l & 15) >> 3))
    b_of = lambda l: 96 * (l & 15) + 64 + 8 * ((l >> 4) ^ (((l & 15) >> 3) << 1))
    for grp in range(4):
        banks = [b for l in range(16 * grp, 16 * grp + 16) for b in range(a_of(l) // 4 % 64, a_of(l) // 4 % 64 + 4)]
        assert len(set(banks)) == 64
    for grp in range(2):
        banks = [b for l in range(32 * grp, 32 * grp + 32) for b in range(b_of(l) // 4 % 64, b_of(l) // 4 % 64 + 2)]
        assert len(set(banks)) == 64
    # ... and that image is the row-major block under the swap: LDS unit u of row r holds block bytes 96 r + 16 (u ^ (r >> 3))
    lds_unit_src = lambda pos: pos ^ (16 if pos // 96 >= 8 else 0)
    for l in range(64):
        r, g = l & 15, l >> 4
        assert lds_unit_src(a_of(l)) == 96 * r + 16 * g and lds_unit_src(b_of(l) & ~15) + (b_of(l) & 15) == 96 * r + 64 + 8 * g
    assert all((pos ^ 16) // 1024 == pos // 1024 for pos in range(0, 3 * 1536, 16))       # a swapped pair never straddles a DMA instruction's KiB


def packed_reference(q, fmt):
    return {1: p16x64_reference, 2: f16x64_reference}[fmt](q)


def packed_unpack(buf, R, KB, fmt):
    if fmt == 3:                                             # F6X128: KB = K / 2 nibble bytes per row of the plain matrix
        return O.pack_i4(f6x128_unpack(buf, R, KB * 2))
    if fmt == 4:                                             # R6X128: the activation side's row-contiguous form
        return O.pack_i4(f6x128_unpack(r6x128_to_fragment_order(buf, R, KB * 2), R, KB * 2))
    return {1: p16x64_unpack, 2: f16x64_unpack}[fmt](buf, R, KB)


@settings(max_examples=25, deadline=None)
@given(st.integers(1, 40), st.integers(1, 3), st.integers(0, 2 ** 31 - 1))
def test_f16x64_layout_is_a_bijection_and_conflict_free(rows, kblocks, seed):
    rng = np.random.default_rng(seed)
    q = rng.integers(-128, 128, (rows, 64 * kblocks), dtype=np.int8)
    buf = f16x64_reference(q)
    assert np.array_equal(f16x64_unpack(buf, rows, 64 * kblocks), q.view(np.uint8))
    # lane l reads bytes [16 l, 16 l + 16) of a block: every 16-lane service group of ds_read_b128 (MI355X_MICROARCH.md, LDS) covers
    # 16 distinct 16-byte slots of the 256-byte bank row, for the 16x16x64 fragment (row l&15, chunk l>>4) ...
    groups = ([0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31],
              [32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59], [36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63])
    for g in groups:
        assert len({l % 16 for l in g}) == 16
    # ... and for the 32x32x32 fragment of the decode kernel (row l&31 over two blocks, chunk 2 sb + (l>>5))
    for sb in range(2):
        for g in groups:
            slots = {(((l & 31) >> 4) * 1024 + (2 * sb + (l >> 5)) * 256 + (l & 15) * 16) // 16 % 16 for l in g}
            assert len(slots) == 16


@settings(max_examples=50, deadline=None)
@given(st.integers(1, 9), st.integers(1, 40), st.integers(0, 2 ** 31 - 1))
def test_pack_unpack_i4_roundtrip(rows, half_cols, seed):
    rng = np.random.default_rng(seed)
    x = rng.integers(-8, 8, (rows, 2 * half_cols), dtype=np.int8)
    p = O.pack_i4(x)
    assert p.shape == (rows, half_cols) and p.dtype == np.uint8
    assert np.array_equal(O.unpack_i4_all(p), x)
    assert np.array_equal(pack_to_i4(torch.from_numpy(x)).numpy(), p)
    # low nibble = even column, two's complement (linear.py:12-18)
    assert np.array_equal(p & 0xF, np.where(x[:, 0::2] < 0, x[:, 0::2] + 16, x[:, 0::2]).astype(np.uint8))
    assert np.array_equal(p >> 4, np.where(x[:, 1::2] < 0, x[:, 1::2] + 16, x[:, 1::2]).astype(np.uint8))


@settings(max_examples=25, deadline=None)
@given(st.integers(1, 40), st.integers(1, 3), st.integers(0, 2 ** 31 - 1))
def test_p16x64_layout_is_a_bijection(rows, kblocks, seed):
    rng = np.random.default_rng(seed)
    q = rng.integers(-128, 128, (rows, 64 * kblocks), dtype=np.int8)
    buf = p16x64_reference(q)
    assert np.array_equal(p16x64_unpack(buf, rows, 64 * kblocks), q.view(np.uint8))
    # every 16-lane group of the GEMM's fragment reads (rows r..r+15 step pattern of ds_read_b128, same logical chunk)
    # lands on 16 distinct 16-byte bank slots of the 256-byte LDS bank row
    for group in ([0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]):
        for c in range(4):
            slots = {((r * 64 + ((c ^ ((-(r >> 2)) & 3)) * 16)) // 16) % 16 for r in group}
            assert len(slots) == 16
    # ... and so does the 16x16x64 fragment of gemm_wreg.hip: lane l reads row l & 15, chunk l >> 4 of ONE block
    for group in ([0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31],
                  [32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59], [36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63]):
        slots = {(((l & 15) * 64 + (((l >> 4) ^ ((-((l & 15) >> 2)) & 3)) * 16)) // 16) % 16 for l in group}
        assert len(slots) == 16


@settings(max_examples=40, deadline=None)
@given(st.integers(1, 6), st.sampled_from([16, 64, 200]), st.sampled_from([4, 8]), st.integers(0, 2 ** 31 - 1))
def test_row_quantisation_invariants(rows, K, bit, seed):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((rows, K)) * rng.choice([1e-3, 1.0, 50.0])).astype(np.float16)
    x[0, :] = 0                                               # all-zero row -> scale 0, q 0
    q, s = O.find_row_scale(x, bit)
    qmax = 2 ** (bit - 1) - 1
    qi = q if bit == 8 else O.unpack_i4_all(q)
    assert qi.min() >= -qmax and qi.max() <= qmax
    assert s[0] == 0 and not qi[0].any()
    amax = np.abs(x.astype(np.float32)).max(axis=1)
    assert np.array_equal(s.view(np.uint16), (amax / np.float32(qmax)).astype(np.float16).view(np.uint16))
    sf = s.astype(np.float32)
    nz = sf > 0
    # dequantised value is within half a step (plus fp16 rounding of the scale) of the input
    err = np.abs(qi[nz].astype(np.float32) * sf[nz, None] - x[nz].astype(np.float32))
    assert (err <= 0.5 * sf[nz, None] * 1.01 + 1e-6).all()
    # the element of largest magnitude maps to +-qmax
    assert (np.abs(qi[nz]).max(axis=1) == qmax).all()


def test_host_unpack_inverts_both_packed_layouts():
    """mixq_amd.linear._unpack_host (state_dict of a compacted layer that was moved to the CPU: ADVICE r02) against the reference
    index maps of this file, both formats, ragged row counts."""
    import torch
    from mixq_amd.linear import _unpack_host
    rng = np.random.default_rng(3)
    for R, KB in [(16, 64), (37, 128), (100, 256)]:
        q = rng.integers(-128, 128, size=(R, KB), dtype=np.int8)
        for fmt in (1, 2):
            img = packed_reference(q, fmt)
            back = _unpack_host(torch.from_numpy(img.reshape(-1, KB)), R, fmt).numpy()
            assert np.array_equal(back.view(np.int8), q), (R, KB, fmt)
