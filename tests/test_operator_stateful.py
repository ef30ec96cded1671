"""Property test of the operator's HOST state machine: random sequences of what a serving process does to a Llama MLP
block - forwards at changing batch sizes, inputs that grow new outlier columns, load_state_dict, .half(), deepcopy, state_dict reads,
the joint gate / up route switched on and off, a layer called on its own - against a TWIN of the same block that keeps nothing between
forwards (every caching feature of MixqConfig off: no kept argument blocks, no joint image, no compaction, no kept outlier maps, no row
maxima handed over).  Invariant after every step: both blocks return the same bytes and hold the same reference-visible state
(`ind`, `weight_cache`, `cnt`, `add_outliers`, `forward_without_precondition_len`, state_dict) - a derived buffer that outlived what it
was made from would show as a different y.  The twin itself is held to the reference by the G5 / G8 fixtures.

Runs the PRODUCT modules (mixq_amd.linear / fused) on tests/backend_oracle.py: the oracle's arithmetic behind the real packed layouts, so
the code under test is the code that runs on the GPU; tests/test_gpu_round6.py has a short GPU twin of one fixed sequence.
Guards /root/reference/mixquant/modules/linear.py:165-289, :292-376 and fused/mlp.py:57-70 semantics under the caching this repo added."""
import copy

import numpy as np
import pytest
import torch
from conftest import swap_backend
from hypothesis import HealthCheck, settings, strategies as st
from hypothesis.stateful import RuleBasedStateMachine, initialize, invariant, precondition, rule

import backend_oracle
import mixq_amd.fused as F
import mixq_amd.linear as L
from mixq_amd import FasterTransformerRMSNorm, MixLibCache, MixLinear_GEMM, MixLlamaMLP, MixqConfig

PLAIN = dict(compact_weights=False, one_call_forward=False, fuse_down_amax=False, joint_gate_up=False, norm_kept_map=False)


def _bits(t):
    return t.detach().contiguous().view(torch.int16) if t.dtype == torch.float16 else t.detach()


def make_block(bit, K, I, seed, config, fp=32, dev="cpu"):
    g = torch.Generator().manual_seed(seed)
    mkw = lambda o, i: (torch.randn(o, i, generator=g) / (i ** 0.5)).half()
    lins = []
    for o, i in ((I, K), (I, K), (K, I)):
        l = torch.nn.Linear(i, o, bias=False).half()
        l.weight.data.copy_(mkw(o, i))
        lins.append(l)
    up, gate, down = lins
    cache = MixLibCache(32, bit=bit, device=dev, config=config)
    ls = torch.rand(K, generator=torch.Generator().manual_seed(seed + 1)) + 0.1 if bit == 4 else None
    mk = lambda l, b, s: MixLinear_GEMM.from_linear(l, b, cache=cache, layer_scales=s, dev=dev, fp_features_num=fp)
    up_q, gate_q, down_q = mk(up, bit, ls), mk(gate, bit, ls), mk(down, 8, None)
    norm = FasterTransformerRMSNorm((1 + 0.1 * torch.randn(K, generator=g)).half().to(dev), eps=1e-6, cache=cache)
    block = MixLlamaMLP(gate_q, down_q, up_q, cache)
    norm.next_layer = block.up_proj_
    return torch.nn.ModuleList([norm, block])              # (one container: deepcopy keeps the cache shared between norm, block and layers)


class BlockMachine(RuleBasedStateMachine):
    BIT, K, I = 8, 128, 64
    DEV, BACKEND = "cpu", backend_oracle                   # (tests/test_gpu_round6.py: "cuda" and the product's own HIP backend)

    def __init__(self):
        super().__init__()
        self._prev = (swap_backend(L, self.BACKEND), swap_backend(F, self.BACKEND)) if self.BACKEND is not None else None
        self.a = self.b = None

    def teardown(self):
        if self._prev is not None:
            swap_backend(L, self._prev[0])
            swap_backend(F, self._prev[1])

    @initialize(seed=st.integers(0, 3))
    def build(self, seed):
        self.a = make_block(self.BIT, self.K, self.I, seed, MixqConfig(), dev=self.DEV)   # every caching feature on (the defaults)
        self.b = make_block(self.BIT, self.K, self.I, seed, MixqConfig(**PLAIN), dev=self.DEV)   # the twin: nothing kept between forwards
        self.hot = [5, 77]
        self.step = 0

    # ---- what a serving process does ------------------------------------------------------------------------------------------------
    def _x(self, M, seed):
        x = torch.randn(M, self.K, generator=torch.Generator().manual_seed(1000 + seed)).half()
        x[:, self.hot] *= 30
        return x.to(self.DEV)

    @rule(M=st.sampled_from([1, 3, 8, 16, 24]), seed=st.integers(0, 50), three_d=st.booleans())
    def forward(self, M, seed, three_d):
        self.step += 1
        x = self._x(M, seed)
        if three_d and M % 2 == 0:
            x = x.reshape(2, M // 2, self.K)
        ya = self.a[1](self.a[0](x.clone()))
        yb = self.b[1](self.b[0](x.clone()))
        assert ya.shape == yb.shape and torch.equal(_bits(ya), _bits(yb)), f"step {self.step}: block output differs from the cache-free twin's"

    @rule(col=st.integers(0, 127))
    def new_hot_column(self, col):
        if col not in self.hot and len(self.hot) < 6:
            self.hot.append(col)

    @rule(M=st.sampled_from([2, 8, 16]), seed=st.integers(0, 50), which=st.sampled_from(["up_proj_", "down_proj_"]))
    def layer_on_its_own(self, M, seed, which):
        """o_proj-style direct call (unfused=True) of a layer that may have given its weights to the joint image"""
        la, lb = getattr(self.a[1], which), getattr(self.b[1], which)
        x = torch.randn(M, la.in_features, generator=torch.Generator().manual_seed(2000 + seed)).half().to(self.DEV)
        ya, yb = la(x.clone(), None, True), lb(x.clone(), None, True)
        assert torch.equal(_bits(ya), _bits(yb))

    @rule(seed=st.integers(0, 3))
    def load_new_weights(self, seed):
        """a checkpoint load into the live block (both blocks get the same tensors)"""
        src = make_block(self.BIT, self.K, self.I, 100 + seed, MixqConfig(**PLAIN), dev=self.DEV)
        sd = src.state_dict()
        # (a 4-bit layer that found new outlier columns has GROWN its registered `ind` / `weight_cache` buffers - linear.py:217-219 assigns the
        # hstack to the buffer names - and a fresh checkpoint no longer fits: the reference fails the same way; both blocks must agree)
        errs = []
        for blk in (self.a, self.b):
            try:
                blk.load_state_dict(copy.deepcopy(sd))
                errs.append(None)
            except RuntimeError as e:
                errs.append("size mismatch" in str(e))
        assert errs[0] == errs[1], errs
        if errs[0] is not None:                               # a refused load leaves both blocks as they were?  the next steps tell
            assert errs[0] is True

    @rule()
    def reload_own_state_dict(self):
        self.a.load_state_dict(self.a.state_dict())

    @rule()
    def half(self):
        self.a.half()                                         # (moves nothing: every tensor is fp16 / integer already)

    @rule()
    def deepcopy(self):
        self.a = copy.deepcopy(self.a)

    @rule()
    def toggle_joint_route(self):
        cfg = self.a[1].config
        cfg.joint_gate_up = not cfg.joint_gate_up

    @rule()
    def toggle_compaction_and_plans(self):
        cfg = self.a[1].config
        cfg.one_call_forward = not cfg.one_call_forward

    @rule()
    def toggle_small_batch_image(self):
        """4-bit layers: a second, nibble image for batches of at most 8 rows (MixqConfig.small_batch_m4); nothing changes for 8-bit layers"""
        cfg = self.a[1].config
        cfg.small_batch_m4 = 0 if cfg.small_batch_m4 else 8

    # ---- invariants --------------------------------------------------------------------------------------------------------------------
    @invariant()
    def same_reference_visible_state(self):
        if self.a is None:
            return
        for nm in ("up_proj_", "gate_proj_", "down_proj_"):
            la, lb = getattr(self.a[1], nm), getattr(self.b[1], nm)
            assert torch.equal(la.ind, lb.ind), nm
            assert la.cnt == lb.cnt and la.add_outliers == lb.add_outliers and la.forward_without_precondition_len == lb.forward_without_precondition_len, nm
            if lb.weight_cache is not None and lb.ind.numel():
                assert torch.equal(_bits(la.weight_cache), _bits(lb.weight_cache)), nm

    @precondition(lambda self: self.a is not None and self.step % 3 == 0)
    @invariant()
    def same_state_dict(self):
        sa, sb = self.a.state_dict(), self.b.state_dict()
        assert sa.keys() == sb.keys()
        for k in sa:
            assert torch.equal(_bits(sa[k]), _bits(sb[k])), k


class BlockMachine4(BlockMachine):
    BIT, K, I = 4, 256, 64


_SETTINGS = settings(max_examples=60, stateful_step_count=24, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)
TestBlock8 = BlockMachine.TestCase
TestBlock8.settings = _SETTINGS
TestBlock4 = BlockMachine4.TestCase
TestBlock4.settings = _SETTINGS


def fixed_sequence(m):
    try:
        m.build(0)
        m.forward(16, 1, False); m.new_hot_column(9); m.forward(16, 2, True); m.forward(8, 3, False); m.forward(16, 4, False)
        assert m.a[1]._joint is not None and m.b[1]._joint is None            # the block under test took the joint route, the twin never does
        assert m.a[1].down_proj_._d.plan is not None and m.b[1].down_proj_._d.plan is None
        m.same_reference_visible_state(); m.same_state_dict()
        m.layer_on_its_own(8, 5, "up_proj_"); m.forward(16, 6, False)
        m.reload_own_state_dict(); m.forward(16, 7, False); m.deepcopy(); m.forward(24, 8, True)
        m.toggle_joint_route(); m.forward(16, 9, False); m.toggle_joint_route(); m.forward(16, 10, False)
        m.load_new_weights(1); m.forward(16, 11, False); m.half(); m.forward(3, 12, False)
        m.same_reference_visible_state(); m.same_state_dict()
    finally:
        m.teardown()


def test_fixed_sequence_on_the_oracle_backend():
    """One hand-written sequence through the same machine (what tests/test_gpu_round6.py replays on the GPU): search, a new column, freeze,
    joint route, a direct layer call, state_dict round trip, deepcopy, new weights."""
    fixed_sequence(BlockMachine())
    fixed_sequence(BlockMachine4())
