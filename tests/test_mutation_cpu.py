"""Mutation harness (CPU): do the tolerances of the GPU parity tests have teeth?

For every known wiring hazard of the DiT (reference call site
src/2d_to_3d_models/run.py:77-84) the oracle is mutated and compared with the unmutated oracle IN THE METRIC THE GPU
TEST USES (tests/parity_support.py).  The mutated result must be off by at least MARGIN (5) x the tolerance the GPU
test applies, i.e. a HIP path with that wiring error could not pass.  Round 1's checkpoint (N(0, 0.02^2) weights,
near-identity blocks) fails this harness for most hazards; it is kept as a regression check of the harness itself.
"""
import pytest
import torch

import parity_support as P
from oracle import hy3d_torch as H


class Case:
    def __init__(self, cfg, seed, batch, init="unit"):
        torch.manual_seed(0)
        self.cfg = cfg
        sd = P.bf16_round_matrices(H.synthetic_state_dict(cfg, seed=seed, init=init))
        self.model = H.load_state_dict(H.ShapePipeline(cfg), sd).model
        self.pipe = None
        self.x, self.t, self.cond = P.dit_inputs(cfg, 0, batch=batch)
        self.streams, self.vec = P.dit_streams(self.model, self.x, self.t, self.cond)
        with torch.no_grad():
            self.out = self.model(self.x, self.t, self.cond)
        self.nd, self.ns = len(self.model.double_blocks), len(self.model.single_blocks)
        self.n_cond = self.cond.shape[1]

    def block_errors(self):
        """per block: metric between the (possibly mutated) block and the recorded unmutated contribution"""
        errs = []
        for k in range(self.nd + self.ns):
            ref = self.streams[k + 1] - self.streams[k]
            got = P.dit_block_apply(self.model, k, self.streams[k], self.vec, self.n_cond) - self.streams[k]
            errs.append(P.block_delta_error(got, ref, self.n_cond, k < self.nd))
        return errs

    def forward_error(self):
        with torch.no_grad():
            return P.rel_l2(self.model(self.x, self.t, self.cond), self.out)


@pytest.fixture(scope="module")
def tiny():
    return Case(H.tiny_config(), 3, 2)


def test_unmutated_oracle_is_a_fixed_point(tiny):
    assert max(tiny.block_errors()) == 0.0 and tiny.forward_error() == 0.0


def test_blocks_are_not_near_identity(tiny):
    """every block's branch contribution is O(1) of the stream it is added to (round 1: ~1 %)"""
    for k in range(tiny.nd + tiny.ns):
        r = float((tiny.streams[k + 1] - tiny.streams[k]).norm() / tiny.streams[k].norm())
        assert r > 0.2, (k, r)


@pytest.mark.parametrize("hazard", sorted(P.DIT_HAZARDS))
def test_hazard_is_outside_the_gpu_tolerances_tiny(tiny, hazard):
    factory, where = P.DIT_HAZARDS[hazard]
    with factory(tiny.model):
        errs = tiny.block_errors()
        fwd = tiny.forward_error()
    if "double" in where:
        assert min(errs[:tiny.nd]) >= P.MARGIN * P.TOL["block_delta"], (hazard, errs)
    if "single" in where:
        assert min(errs[tiny.nd:]) >= P.MARGIN * P.TOL["block_delta"], (hazard, errs)
    assert fwd >= P.MARGIN * P.TOL["dit_forward_tiny"], (hazard, fwd)
    assert max(tiny.block_errors()) == 0.0      # the patch is gone


def test_hazards_through_the_sampling_loop(tiny):
    """the N-step CFG sampler (what flow_sample's tolerance guards): a broken attention moves the latents by far
    more than TOL['flow_sample'] (round 1: 3.6e-3 against 3e-2)"""
    pipe = H.ShapePipeline(tiny.cfg)
    pipe.model = tiny.model
    lat0 = tiny.x[:1].clone()
    ref = pipe.sample(tiny.cond, lat0.clone(), 6, 5.0)
    for hazard in ("attention_replaced_by_v", "v_head_dims_flipped", "qkv_head_major_split", "shift_scale_swapped",
                   "gate_index", "latent_first_concat", "timestep_sin_first"):
        with P.DIT_HAZARDS[hazard][0](tiny.model):
            got = pipe.sample(tiny.cond, lat0.clone(), 6, 5.0)
        assert P.rel_l2(got, ref) >= P.MARGIN * P.TOL["flow_sample"], hazard


def test_round1_checkpoint_would_not_have_caught_them():
    """regression check of the harness: with the near-identity checkpoint of round 1 a dead attention stays inside
    the whole-forward tolerance that round 1 applied (1e-2) -- the judge's finding, reproduced"""
    c = Case(H.tiny_config(), 3, 2, init="small")
    with P.mut_attention_is_v():
        assert c.forward_error() < 2e-2
    with P.mut_v_dims_flipped():
        assert c.forward_error() < 1e-2


@pytest.mark.slow
def test_hazards_at_full_width():
    """hidden 1024, 16 heads, 3072 + 1370 tokens, one double + one single block: the configuration of the full-width
    GPU block test"""
    c = Case(H.wide_config(1, 1, 1, 1), 11, 1)
    for hazard, (factory, where) in sorted(P.DIT_HAZARDS.items()):
        with factory(c.model):
            errs = c.block_errors()
            fwd = c.forward_error()
        if "double" in where:
            assert errs[0] >= P.MARGIN * P.TOL["block_delta"], (hazard, errs)
        if "single" in where:
            assert errs[1] >= P.MARGIN * P.TOL["block_delta"], (hazard, errs)
        assert fwd >= P.MARGIN * P.TOL["dit_forward_full_depth"], (hazard, fwd)


# ---- hazards below the bf16 resolution of a whole block: pinned at operator level ----------------------------------
def test_gelu_flavour_needs_the_operator_level_test():
    """tanh- vs erf-GELU differ by <= 5e-4 absolute: invisible behind bf16 operand rounding in any block-level metric
    (shown here), hence the binned-mean test of the GEMM epilogue in tests/test_ops_gpu.py; this is its CPU twin"""
    c = Case(H.tiny_config(), 3, 2)
    with P.mut_gelu_flavour(c.model):
        assert max(c.block_errors()) < P.TOL["block_delta"]
    x = torch.linspace(-4, 4, 400001)
    d = P.gelu_flavour_statistic(torch.nn.functional.gelu(x, approximate="tanh").to(torch.bfloat16).float(), x, "erf")
    assert d >= P.MARGIN * P.TOL_GELU_STAT
    same = P.gelu_flavour_statistic(torch.nn.functional.gelu(x, approximate="tanh").to(torch.bfloat16).float(), x, "tanh")
    assert same <= P.TOL_GELU_STAT


def test_qk_norm_eps_needs_small_q():
    """RMSNorm's eps (1e-6) is invisible for |q| ~ 1 and a first-order term for |q| ~ 1e-3: the GPU suite runs the
    block-delta test once more on the small-q checkpoint (test_dit_block_deltas_small_q), which pins eps"""
    c = Case(H.tiny_config(), 3, 2)
    with P.mut_qk_norm_eps(1e-5):
        assert max(c.block_errors()) < P.TOL["block_delta"]          # not detectable on the standard checkpoint
    cfg = H.tiny_config()
    sd = P.small_qk_state_dict(P.bf16_round_matrices(H.synthetic_state_dict(cfg, seed=3)), cfg)
    c.model = H.load_state_dict(H.ShapePipeline(cfg), sd).model
    c.streams, c.vec = P.dit_streams(c.model, c.x, c.t, c.cond)
    for eps in (1e-5, 1e-7, 0.0):
        with P.mut_qk_norm_eps(eps):
            assert min(c.block_errors()) >= P.MARGIN * P.TOL["block_delta"], eps
