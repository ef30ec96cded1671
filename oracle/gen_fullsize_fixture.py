#!/usr/bin/env python3
"""tests/golden/g7_fullsize_512x4096x11008.npz: the ORACLE's outputs at the metric shape (BASELINE.json configs[1]: 512 tokens, Llama-2-7b
up_proj 4096 -> 11008, W8A8O16, 41 outlier columns x 20), so that the GPU box can check the full-size operator against committed numbers
without running the oracle.  TEST INFRASTRUCTURE ONLY.  Everything is regenerated from seeds by the test; the
fixture holds what the oracle computed from them:
    ind [41] int32 (sorted), rows [8], q_x[rows] int8 [8, 4096], x_scale fp16 [512] (every row), x_out[rows] fp16 [8, 41],
    y[rows] fp16 [8, 11008], scale_col fp16 [11008], q_weight row sums int32 [11008] (pins the weight quantisation), weight_cache[:, :4]
Run here (CPU, ~20 s):  python oracle/gen_fullsize_fixture.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[0] = ROOT
from oracle import oracle as O  # noqa: E402

M, K, N = 512, 4096, 11008
ROWS = [0, 1, 63, 137, 255, 256, 300, 511]


def inputs():
    """The seeded problem, as the test rebuilds it: nn.Linear default init (seed 0), activations N(0,1) fp16 (seed 12), 41 columns x 20."""
    torch.manual_seed(0)
    lin = torch.nn.Linear(K, N, bias=False).half()
    cols = torch.randperm(K, generator=torch.Generator().manual_seed(1))[: round(0.01 * K)]
    x = torch.randn(M, K, generator=torch.Generator().manual_seed(12)).half()
    x[:, cols] *= 20
    ind = np.sort(cols.numpy()).astype(np.int32)
    return lin.weight.detach().numpy(), x.numpy(), ind


def main():
    w, x, ind = inputs()
    qw, sw = O.quant_weight_w8(w)                                   # linear.py:111-119
    wo = O.dequant_weight_cols(qw, sw, ind, 8)                      # linear.py:207
    xz = x.copy()
    xo = O.extract_outliers_zero(xz, ind)                           # linear.py:187-193 (k2)
    qx, sx = O.find_row_scale(xz, 8)                                # ... (k1)
    y = O.linear_fused(qx[ROWS], qw, sx[ROWS], sw, xo=xo[ROWS], wo=wo, bit=8)
    out = os.path.join(ROOT, "tests", "golden", "g7_fullsize_512x4096x11008.npz")
    np.savez_compressed(out, ind=ind, rows=np.array(ROWS, np.int32), q_x=qx[ROWS], x_scale=sx, x_out=xo[ROWS], y=y.astype(np.float16),
                        scale_col=sw.reshape(-1), q_weight_rowsum=qw.astype(np.int32).sum(axis=1).astype(np.int32), weight_cache_head=wo[:, :4],
                        x_zeroed_colsum=np.abs(xz[:, ind].astype(np.float32)).sum())
    print(out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
