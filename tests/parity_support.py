"""Shared by the GPU parity tests (tests/test_model_gpu.py) and the CPU mutation harness
(tests/test_mutation_cpu.py): the stated tolerances, the metrics, and the catalogue of wiring hazards.

The DiT / VAE oracle (oracle/hy3d_torch.py) is a self-written restatement ("parity unpinned"); what these tests CAN
establish is that the HIP path computes the same function as that restatement, with tolerances tight enough that
every known wiring hazard would be caught.  The mutation harness proves the second half: for each hazard, the
mutated oracle differs from the unmutated one, in the metric the GPU test uses, by at least MARGIN x the tolerance
the GPU test applies.
"""
import contextlib

import torch

# ---- tolerances used by the -m gpu tests (bf16 GEMM operands / fp32 accumulation against the fp32 oracle) ----------
TOL = {
    # |delta_gpu - delta_oracle| / |delta_oracle| for ONE block applied to the SAME input stream (delta = block(x) - x)
    # (measured on MI355X, round 2: 2.1e-3 .. 2.9e-3 for all 48 full-width blocks)
    "block_delta": 5e-3,
    # whole DiT forward (velocity), rel-L2 (measured 2.1e-3 .. 3.0e-3, full depth 3.0e-3)
    "dit_forward_tiny": 8e-3,
    "dit_forward_full_depth": 1e-2,
    # N-step CFG sampling (latents), rel-L2: guidance 5 amplifies the per-step error (measured 7.5e-3 for 6 steps)
    "flow_sample": 2e-2,
    # 50 steps x CFG 2, the reference's setting: SURVEY 8(c) states 3e-2 for the 50-step latents
    "flow_sample_50": 3e-2,
    # shape-VAE transformer output, rel-L2 (measured 2.5e-3); grid logits: max |d| / max |logit| (measured 3.2e-3)
    "vae_latents": 8e-3,
    "grid_logits": 1e-2, "grid_logits_fp8": 6e-2,
    "conditioner": 1e-2,      # measured 3.9e-3
    # two valid bf16 evaluations of the same sampler (different tile partitions) against each other
    "same_function": 2e-2,
}
MARGIN = 5.0   # a hazard counts as detectable when it moves the metric by >= MARGIN x the tolerance

MEASURED = []  # (name, value, tolerance) collected by the GPU tests; printed in the terminal summary (conftest.py)


def report(name, value, tol):
    MEASURED.append((name, float(value), float(tol)))


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float(torch.linalg.norm(a - b) / (torch.linalg.norm(b) + 1e-30))


def dit_inputs(cfg, seed=0, batch=2, uncond_zero=True):
    g = torch.Generator().manual_seed(seed)
    d = cfg["dit"]
    Lc = (cfg["cond"]["image_size"] // cfg["cond"]["patch_size"]) ** 2 + 1
    x = torch.randn(batch, cfg["vae"]["num_latents"], d["in_channels"], generator=g)
    cond = torch.randn(batch, Lc, d["context_in_dim"], generator=g).to(torch.bfloat16).float()
    if uncond_zero and batch == 2:
        cond[1] = 0
    t = torch.full((batch,), 0.37)
    return x, t, cond


def bf16_round_matrices(sd):
    """the checkpoint as both sides see it: matrices representable in bf16 (rounded once), vectors fp32"""
    out = {}
    for k, v in sd.items():
        if torch.is_floating_point(v) and v.ndim >= 2 and not k.endswith(("cls_token", "mask_token", "position_embeddings",
                                                                         "output_proj.weight")):
            out[k] = v.to(torch.bfloat16).to(torch.float32)
        else:
            out[k] = v.clone()
    return out


# ---- the DiT unrolled block by block (same arithmetic as Hunyuan3DDiT.forward) --------------------------------------
@torch.no_grad()
def dit_prologue(model, x, t, cond):
    from oracle import hy3d_torch as H
    latent = model.latent_in(x)
    vec = model.time_in(H.timestep_embedding(t, 256, time_factor=model.time_factor).to(latent.dtype))
    return latent, model.cond_in(cond), vec


@torch.no_grad()
def dit_block_apply(model, k, stream, vec, n_cond):
    """stream = joint residual [B, n_cond + n_lat, H] in upstream order (cond first) BEFORE block k (double blocks first,
    then single blocks) -> the stream after it."""
    from oracle import hy3d_torch as H
    nd = len(model.double_blocks)
    if k < nd:
        cond, latent = H._unjoint(stream, n_cond)
        latent, cond = model.double_blocks[k](latent, cond, vec)
        return H._joint(cond, latent)
    return model.single_blocks[k - nd](stream, vec)


@torch.no_grad()
def dit_streams(model, x, t, cond):
    """[stream_0 (after latent_in / cond_in), stream_1 (after block 0), ...], vec"""
    from oracle import hy3d_torch as H
    latent, c, vec = dit_prologue(model, x, t, cond)
    s = H._joint(c, latent)
    out = [s]
    for k in range(len(model.double_blocks) + len(model.single_blocks)):
        s = dit_block_apply(model, k, s, vec, c.shape[1])
        out.append(s)
    return out, vec


def block_delta_error(got, ref, n_cond, is_double):
    """the metric of the per-block tests: rel-L2 of the branch contribution; for a double block the worse of its two
    streams (the short conditioning stream must not hide behind the long latent one)"""
    if is_double:
        return max(rel_l2(got[:, :n_cond], ref[:, :n_cond]), rel_l2(got[:, n_cond:], ref[:, n_cond:]))
    return rel_l2(got, ref)


# ---- hazards -----------------------------------------------------------------------------------------------------------
@contextlib.contextmanager
def _patched(obj, name, new):
    old = getattr(obj, name)
    setattr(obj, name, new)
    try:
        yield
    finally:
        setattr(obj, name, old)


def _H():
    from oracle import hy3d_torch as H
    return H


def mut_qkv_head_major():
    """fused qkv read as "(H K D)" (per-head interleaved, the ShapeVAE convention) instead of "(K H D)" """
    def split(qkv, heads):
        B, L, _ = qkv.shape
        return qkv.view(B, L, heads, 3, -1).permute(3, 0, 2, 1, 4)
    return _patched(_H(), "_split_khd", split)


def mut_latent_first_concat():
    """cat(latent, cond) while the outputs are still split as [cond | latent]"""
    return _patched(_H(), "_joint", lambda txt, img: torch.cat((img, txt), dim=-2))


def mut_shift_scale_swapped():
    return _patched(_H(), "_modulate", lambda x, shift, scale: (1 + shift) * x + scale)


def mut_gate_index():
    """the attention branch gated by gate2 and the MLP branch by gate1 (double); gate <-> shift (single)"""
    H = _H()
    orig = H.Modulation.forward

    def fwd(self, vec):
        a, b = orig(self, vec)
        if b is None:
            return (a[2], a[1], a[0]), None
        return (a[0], a[1], b[2]), (b[0], b[1], a[2])
    return _patched(H.Modulation, "forward", fwd)


def mut_mod_chunk_order():
    """modulation output chunked (scale, shift, gate) instead of (shift, scale, gate)"""
    H = _H()
    orig = H.Modulation.forward

    def fwd(self, vec):
        a, b = orig(self, vec)
        a = (a[1], a[0], a[2])
        if b is not None:
            b = (b[1], b[0], b[2])
        return a, b
    return _patched(H.Modulation, "forward", fwd)


@contextlib.contextmanager
def mut_gelu_flavour(model):
    """GELU(tanh) <-> exact (erf) in every MLP of `model`"""
    import torch.nn as nn
    gelus = [m for m in model.modules() if isinstance(m, nn.GELU)]   # nn.GELU.forward reads self.approximate per call
    for m in gelus:
        m.approximate = "none" if m.approximate == "tanh" else "tanh"
    try:
        yield
    finally:
        for m in gelus:
            m.approximate = "none" if m.approximate == "tanh" else "tanh"


def mut_v_dims_flipped():
    H = _H()
    return _patched(H, "_sdpa", lambda q, k, v: torch.nn.functional.scaled_dot_product_attention(q, k, v.flip(-1)))


def mut_attention_is_v():
    """attention replaced by its value input (what a dead softmax / missing attention would give)"""
    def f(q, k, v):
        L = q.shape[-2]
        return v[..., :L, :] if v.shape[-2] >= L else v.mean(-2, keepdim=True).expand(*q.shape[:-1], v.shape[-1])
    return _patched(_H(), "_sdpa", f)


def mut_no_softmax_scale():
    """scores scaled by 1 instead of 1/sqrt(64)"""
    return _patched(_H(), "_sdpa", lambda q, k, v: torch.nn.functional.scaled_dot_product_attention(q, k, v, scale=1.0))


def mut_qk_norm_dropped():
    H = _H()
    return _patched(H.QKNorm, "forward", lambda self, q, k, v: (q, k))


def mut_qk_norm_eps(eps=1e-2):
    """RMSNorm eps; only visible when q / k are small (see the small-q checkpoint of the op-level test)"""
    return _patched(_H().RMSNorm, "eps", eps)


def mut_timestep_sin_first():
    H = _H()
    orig = H.timestep_embedding

    def emb(t, dim, max_period=10000, time_factor=1000.0):
        e = orig(t, dim, max_period, time_factor)
        return torch.cat([e[:, dim // 2:], e[:, :dim // 2]], dim=-1)
    return _patched(H, "timestep_embedding", emb)


def mut_time_factor_one():
    H = _H()
    orig = H.timestep_embedding
    return _patched(H, "timestep_embedding", lambda t, dim, max_period=10000, time_factor=1000.0: orig(t, dim, max_period, 1.0))


# name -> (factory taking the oracle model, which blocks it must be visible in: "double", "single", "forward")
DIT_HAZARDS = {
    "qkv_head_major_split": (lambda m: mut_qkv_head_major(), ("double", "single")),
    "latent_first_concat": (lambda m: mut_latent_first_concat(), ("double", "forward")),
    "shift_scale_swapped": (lambda m: mut_shift_scale_swapped(), ("double", "single")),
    "gate_index": (lambda m: mut_gate_index(), ("double", "single")),
    "modulation_chunk_order": (lambda m: mut_mod_chunk_order(), ("double", "single")),
    "v_head_dims_flipped": (lambda m: mut_v_dims_flipped(), ("double", "single")),
    "attention_replaced_by_v": (lambda m: mut_attention_is_v(), ("double", "single")),
    "softmax_scale_missing": (lambda m: mut_no_softmax_scale(), ("double", "single")),
    "qk_norm_dropped": (lambda m: mut_qk_norm_dropped(), ("double", "single")),
    "timestep_sin_first": (lambda m: mut_timestep_sin_first(), ("forward",)),
    "time_factor_missing": (lambda m: mut_time_factor_one(), ("forward",)),
}


# ---- GELU flavour, statistically (bf16 outputs hide it element by element) ----------------------------------------------
TOL_GELU_STAT = 5e-5


def gelu_flavour_statistic(y, x, flavour):
    """max over bins of x in [-3.5, -2) of |mean(y - gelu_flavour(x))|.  The tanh form deviates from the exact one by an
    EVEN function of x that peaks at 4.7e-4 near |x| = 2.75; on the negative side gelu(x) ~ -0.008, where the bf16 output
    step is 6e-5 and its rounding noise averages out over a bin -- on the positive side (values ~ 2.7, step 0.016) it
    would not.  x, y: fp32 tensors of equal shape."""
    ref = torch.nn.functional.gelu(x.double(), approximate="tanh" if flavour == "tanh" else "none")
    err = (y.double() - ref).reshape(-1)
    xb = x.reshape(-1)
    worst, used = 0.0, 0
    edges = torch.linspace(-3.5, -2.0, 7)
    for lo, hi in zip(edges[:-1], edges[1:]):
        sel = (xb >= lo) & (xb < hi)
        if int(sel.sum()) >= 500:
            used += 1
            worst = max(worst, abs(float(err[sel].mean())))
    assert used >= 4, "too few samples in the bins"
    return worst


# ---- RMSNorm eps of the q/k norm: only visible when q, k are tiny ---------------------------------------------------------
def small_qk_state_dict(sd, cfg, factor=2.0 ** -10):   # a power of two: the matrices stay bf16-representable
    """the same checkpoint with the q and k rows of every fused DiT projection (weights and biases) scaled by `factor`:
    mean(q^2) drops to ~1e-6, the size of RMSNorm's eps, so eps becomes a first-order term of the normalisation
    (query-norm scales x3 keep the softmax sharp enough for the attention output to depend on it)"""
    Hd = cfg["dit"]["hidden_size"]
    out = {k: v.clone() for k, v in sd.items()}
    for k in out:
        if k.startswith("model.") and (k.endswith("_attn.qkv.weight") or k.endswith("_attn.qkv.bias")
                                       or k.endswith(".linear1.weight") or k.endswith(".linear1.bias")):
            out[k][:2 * Hd] *= factor
        if k.startswith("model.") and k.endswith("query_norm.scale"):
            out[k] *= 3.0
    return out
