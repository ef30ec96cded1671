#!/usr/bin/env python3
"""Debug aid: one forced GEMM configuration on one shape, checked against torch's integer product (exit code 0 / 1), own process
per case so a faulting kernel does not take the other cases with it."""
import os, sys
os.environ.setdefault("MIXQ_TUNING_LIB", "1")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixq_amd import _capi, mixlib
cfg, shp, reps = sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 1
M, N, K = (int(v) for v in shp.split("x"))
lib = _capi.load()
names = _capi.gemm_config_names()
g = torch.Generator().manual_seed(0)
qx = torch.randint(-127, 128, (M, K), generator=g, dtype=torch.int8).cuda()
qw = torch.randint(-128, 128, (N, K), generator=g, dtype=torch.int8).cuda()
sx = torch.full((M, 1), 2.0 ** -7, dtype=torch.float16, device="cuda")
sw = torch.full((1, N), 2.0 ** -7, dtype=torch.float16, device="cuda")
want = ((qx.double() @ qw.double().T) * 2.0 ** -14).to(torch.float16)
xp, wp = mixlib.PackOperand(qx, 1), mixlib.PackOperand(qw, 2)
assert lib.mixq_gemm_set_config(names.index(cfg)) == 0
out = torch.zeros((M, N), dtype=torch.float16, device="cuda")
for r in range(reps):
    mixlib.FusedLinear(xp, wp, sx, sw, None, None, 0, None, M, N, K, out=out)
torch.cuda.synchronize()
bad = (out != want)
print(cfg, shp, "reps", reps, "mismatches", int(bad.sum()), "of", M * N, flush=True)
if bad.any():
    idx = bad.nonzero()
    print("  first bad (m, n):", idx[:5].tolist(), " rows hit:", idx[:, 0].unique().numel(), " cols hit:", idx[:, 1].unique().numel())
sys.exit(1 if bad.any() else 0)
