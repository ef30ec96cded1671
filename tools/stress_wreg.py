#!/usr/bin/env python3
"""Stress for the hand-counted weight rings (the hazards there are timing dependent): every weights-in-registers tiling, int8 and FP6 form, replayed
many times inside graphs that interleave several shapes (different co-runners, cold and warm buffers), every replay's output compared bit for bit with
the LDS-staged kernels' result on the same operands.  Exit code 1 on any mismatch."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixq_amd import _capi, mixlib
dev = "cuda"
lib = _capi.load(); names = _capi.gemm_config_names()
g = torch.Generator().manual_seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 30
def nib(R, K, lo):
    v = torch.randint(lo, 8, (R, K), generator=g, dtype=torch.int8); u = (v & 0xF).to(torch.uint8)
    return (u[:, 0::2] | (u[:, 1::2] << 4)).contiguous().to(dev)
cases = []
for (M, N, K, n_out) in [(512, 11008, 4096, 41), (512, 4096, 11008, 110), (200, 328, 512, 19), (512, 14336, 4096, 64), (96, 4096, 1024, 0), (512, 12288, 4096, 128), (130, 6144, 2048, 33)]:
    sx = (torch.rand(M, 1, generator=g) * 0.01 + 0.001).half().to(dev); sw = (torch.rand(1, N, generator=g) * 0.01 + 0.001).half().to(dev)
    xo = wo = None
    if n_out:
        pad = (n_out + 15) // 16 * 16
        xo = torch.randn((M, pad), generator=g).half().to(dev)[:, :n_out]; wo = torch.randn((N, pad), generator=g).half().to(dev)[:, :n_out]
    bias = torch.randn(N, generator=g).half().to(dev)
    q8x = torch.randint(-127, 128, (M, K), generator=g, dtype=torch.int8).to(dev); q8w = torch.randint(-127, 128, (N, K), generator=g, dtype=torch.int8).to(dev)
    q4x, q4w = nib(M, K, -7), nib(N, K, -8)
    ops8 = (mixlib.PackOperand(q8x, 1), mixlib.PackOperand(q8w, 2), mixlib.PackOperand(q8w, 1))
    ops4 = (mixlib.PackOperand(q4x, 4), mixlib.PackOperand(q4w, 3), mixlib.PackOperand(q4x, 1), mixlib.PackOperand(q4w, 1))
    lib.mixq_gemm_set_config(names.index("128x128_w2x2_s5_l2"))
    ref8 = mixlib.FusedLinear(ops8[0], ops8[2], sx, sw, xo, wo, n_out, bias, M, N, K, bit=8)
    ref4 = mixlib.FusedLinear(ops4[2], ops4[3], sx, sw, xo, wo, n_out, bias, M, N, K, bit=4)
    lib.mixq_gemm_set_config(-1)
    torch.cuda.synchronize()
    cases.append(dict(M=M, N=N, K=K, n_out=n_out, sx=sx, sw=sw, xo=xo, wo=wo, bias=bias, ops8=ops8, ops4=ops4, ref8=ref8, ref4=ref4))
wr8 = [nm for nm in names if nm.startswith("wr") and "abl" not in nm and "_p" not in nm and "f6" not in nm and "self" not in nm and not nm.endswith(("_k2", "_pair"))]
wr6 = ["wr128x192_s16_d4_l2", "wr128x128_s16_d4_l2", "wr64x128_s16_d4_l2", "wr64x192_s16_d4_l2", "wr64x256_s16_d4_l2"]
bad = 0
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    for bit, cfgs in ((8, wr8), (4, wr6)):
        for nm in cfgs:
            lib.mixq_gemm_set_config(names.index(nm))
            outs = [torch.zeros((c["M"], c["N"]), dtype=torch.float16, device=dev) for c in cases]
            def run_all():
                for c, o in zip(cases, outs):
                    a, b = (c["ops8"][0], c["ops8"][1]) if bit == 8 else (c["ops4"][0], c["ops4"][1])
                    mixlib.FusedLinear(a, b, c["sx"], c["sw"], c["xo"], c["wo"], c["n_out"], c["bias"], c["M"], c["N"], c["K"], bit=bit, out=o)
            run_all(); torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=side):
                run_all(); run_all()
            torch.cuda.synchronize()
            nbad = 0
            for r in range(rounds):
                for o in outs: o.zero_()
                gr.replay(); torch.cuda.synchronize()
                for c, o in zip(cases, outs):
                    if not torch.equal(o, c["ref8"] if bit == 8 else c["ref4"]): nbad += 1
            lib.mixq_gemm_set_config(-1)
            print(f"bit {bit} {nm}: {rounds} replays x {len(cases)} shapes x 2 launches, mismatching outputs: {nbad}", flush=True)
            bad += nbad
# the paired gate / up forms (MIXQ_ACT_SILU_PAIR: their own instantiations of the hand-counted loop): every case's weights doubled into an
# interleaved image of (the layer, the layer with its rows rolled by 8), reference = the LDS-staged kernels' two launches
from mixq_amd.fused import interleave_pair_rows
pair8 = ["wr128x192_s16_d4_l2", "wr128x256_s16_d3_l2", "wr64x192_s16_d4_l2", "wr64x256_s16_d4_l2", "wr32x64_s8_d6_l1"]
pair6 = ["wr128x192_s16_d4_l2", "wr64x192_s16_d4_l2", "wr64x256_s16_d4_l2", "wr32x64_s8_d6_l1"]
pcases = []
with torch.cuda.stream(side):
    for c in cases:
        M, N, K, n_out = c["M"], c["N"], c["K"], c["n_out"]
        if N % 8:
            continue
        def roll(t):                                                             # rows rolled by 8, in the padded layout the tail takes
            if t is None:
                return None
            pad = (t.shape[1] + 15) // 16 * 16
            buf = torch.zeros((t.shape[0], pad), dtype=t.dtype, device=t.device)
            buf[:, :t.shape[1]] = torch.roll(t, 8, 0)
            return buf[:, :t.shape[1]]
        ent = dict(c=c)
        for bit in (8, 4):
            xq = c["ops8"][0] if bit == 8 else c["ops4"][2]                      # P16X64 activations of the LDS-staged reference
            wst = mixlib.UnpackOperand(c["ops8"][2] if bit == 8 else c["ops4"][3], N)   # the plain rows back
            w2 = torch.roll(wst, 8, 0)
            sw2, wo2, b2 = torch.roll(c["sw"], 8, 1), roll(c["wo"]), torch.roll(c["bias"], 8, 0)
            lib.mixq_gemm_set_config(names.index("128x128_w2x2_s5_l2"))
            up = mixlib.FusedLinear(xq, mixlib.PackOperand(wst, 1), c["sx"], c["sw"], c["xo"], c["wo"], n_out, c["bias"], M, N, K, bit=bit)
            ref = mixlib.FusedLinear(xq, mixlib.PackOperand(w2, 1), c["sx"], sw2, c["xo"], wo2, n_out, b2, M, N, K, bit=bit, act=2, addend=up)
            lib.mixq_gemm_set_config(-1)
            jw = mixlib.PackOperand(interleave_pair_rows(wst, w2), 2 if bit == 8 else 3)
            jsw = interleave_pair_rows(c["sw"].reshape(-1), sw2.reshape(-1)).reshape(1, -1)
            jwo = None
            if n_out:
                pad = (n_out + 15) // 16 * 16
                jwo = torch.zeros((2 * N, pad), dtype=torch.float16, device=dev); jwo[:, :n_out] = interleave_pair_rows(c["wo"], wo2); jwo = jwo[:, :n_out]
            ent[bit] = dict(x=c["ops8"][0] if bit == 8 else c["ops4"][0], jw=jw, jsw=jsw, jwo=jwo, jb=interleave_pair_rows(c["bias"], b2), ref=ref)
        pcases.append(ent)
    torch.cuda.synchronize()
    for bit, cfgs in ((8, pair8), (4, pair6)):
        for nm in cfgs:
            lib.mixq_gemm_set_config(names.index(nm))
            outs = [torch.zeros((e["c"]["M"], e["c"]["N"]), dtype=torch.float16, device=dev) for e in pcases]
            def run_pairs():
                for e, o in zip(pcases, outs):
                    c, d = e["c"], e[bit]
                    mixlib.FusedLinear(d["x"], d["jw"], c["sx"], d["jsw"], c["xo"], d["jwo"], c["n_out"], d["jb"], c["M"], 2 * c["N"], c["K"], bit=bit, act=3, out=o)
            run_pairs(); torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=side):
                run_pairs(); run_pairs()
            torch.cuda.synchronize()
            nbad = 0
            for r in range(rounds):
                for o in outs: o.zero_()
                gr.replay(); torch.cuda.synchronize()
                for e, o in zip(pcases, outs):
                    if not torch.equal(o, e[bit]["ref"]): nbad += 1
            lib.mixq_gemm_set_config(-1)
            print(f"bit {bit} PAIR {nm}: {rounds} replays x {len(pcases)} shapes x 2 launches, mismatching outputs: {nbad}", flush=True)
            bad += nbad
print("TOTAL MISMATCHES", bad)
sys.exit(1 if bad else 0)
