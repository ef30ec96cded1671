"""Per-run scratch/state shared by all MixQ linears of a model — mirror of the reference's MixLibCache
(/root/reference/mixquant/Cache.py:5-25): same constructor, same attribute names, so reference-style model code
(`MixLibCache(inputdim, sigma, bit)`, `cache.x_scale`, `cache.zeros`, `cache.q_xcache`, `cache.activation_outliers`,
`cache.ind`, `cache.new_ind`, `cache.shape`, `cache.stop`) works unchanged.

MI355X differences, all behind the same names:
  * `zeros` is a stride-0 view of one fp16 zero instead of an inputdim x 36864 matrix (the reference reads 11 MB of
    zeros per GEMM as the "addend" when a layer has no outliers; here the GEMM gets a NULL addend instead).
  * `flag` / `scratch(K)`: device words for the misprediction flag and the outlier-column detection, so the check
    `x_scale.max() > sigma/qmax` (linear.py:201) is evaluated on device inside the quantise kernel.
  * `n_dev`: the device-resident outlier count of the layer that filled the cache (kernel.py:108-111 reads its K the same
    way): the quantise kernel and the GEMM tail take the count from there, `ind` / `activation_outliers` are capacities.
  * `config`: the model's MixqConfig (config.py) - packed formats, one-call forward, joint gate / up launch ... - instead of
    process-global switches: two models in one process do not change each other's kernels.
  * the storage format of `q_xcache` (plain / P16x64 / F16x64) is a tag on the tensor itself (mixlib.fmt_of), not a flag
    here: a flag on this shared object would describe whichever layer wrote last.
"""
from __future__ import annotations

import torch

from .config import MixqConfig


class MixLibCache:
    def __init__(self, inputdim=1024, sigma=6, bit=8, eval_ppl=False, locality=False, device="cuda", config=None):
        self.device = device
        self.config = config if config is not None else MixqConfig()    # this model's switches (config.py); every layer built with this cache shares it
        self.inputdim = inputdim
        self.x_scale = torch.zeros((inputdim, 1), dtype=torch.float16, device=device)
        self.sigma = torch.zeros((1, 1), dtype=torch.float16, device=device)
        self.sigma[0] = sigma
        self.sigma_value = float(self.sigma.float().cpu().item())       # fp16-rounded threshold as a host float
        self._zero = torch.zeros((1, 1), dtype=torch.float16, device=device)
        self.zeros = self._zero.expand(inputdim, 12288 * 3)              # same shape as Cache.py:11, no storage
        self.zeros._mixq_all_zero = True                                 # recognised by identity: the GEMM gets a NULL addend
        self.ind = None
        self.new_ind = None
        self.shape = None
        self.activation_outliers = None
        self.q_xcache = None
        self.n_dev = None                    # int32[1] device count that goes with `ind` / `activation_outliers`
        self.is_prefill = False
        self.bit = bit
        self.max_outliers = 256
        self.stop = 2
        self.eval_ppl = eval_ppl
        self.locality = locality
        self.flag = torch.zeros((1,), dtype=torch.int32, device=device)
        self._scratch = {}

    def scratch(self, K):
        """(colflags u8[K], ind_buf i32[K], count i32[1]) for mixq_detect_outlier_cols."""
        s = self._scratch.get(K)
        if s is None:
            s = (torch.empty(K, dtype=torch.uint8, device=self.device),
                 torch.empty(K, dtype=torch.int32, device=self.device),
                 torch.zeros(1, dtype=torch.int32, device=self.device))
            self._scratch[K] = s
        return s

    def do_bench_cudagraph(self, fn):
        """Mirror of Cache.py:26-38: ten warm-up calls of `fn`, then `fn` captured into a graph (a hipGraph here) on the CURRENT stream,
        which must not be the default one; returns the graph (`g.replay()`).  The warm-up calls also let outlier prediction freeze
        (Cache.py:21, `stop = 2`): a layer that is still searching synchronises with the host and cannot be captured."""
        if torch.cuda.current_stream() == torch.cuda.default_stream():
            raise RuntimeError("Cannot capture graph in default stream. Please use side stream in benchmark code.")
        for _ in range(10):
            fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        torch.cuda.synchronize()
        return g

    @property
    def q_xcache_packed(self):
        """True when q_xcache is in a packed (tile-major) layout; derived from the tensor's own tag."""
        from .mixlib import fmt_of
        return self.q_xcache is not None and fmt_of(self.q_xcache) != 0


class MLPCache:
    """Mirror of Cache.py:42-48 (used by the GPT-J MLP wrapper)."""

    def __init__(self, max_batch_size=4096, device="cuda"):
        self.device = device
        self.x_scale = torch.zeros((max_batch_size, 1), dtype=torch.float16, device=device)
        self.ind = None
        self.shape = None
        self.activation_outliers = None
