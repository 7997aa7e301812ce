"""BASELINE.json configs[3] as a MEASURED configuration with a stated acceptance (reference knobs
src/config.yaml:167-169): the geo decoder's GEMMs on e4m3 MFMA at a 513^3 grid.  fp8 rounding moves logits by up to 6e-2 of their
maximum (tests/test_model_gpu.py::test_geo_decoder_fp8_mode), incoherently from grid point to grid point, and near the iso-surface
a sign flip legitimately changes the marching-cubes topology -- so the acceptance is a MESH property, stated on the
surface extracted from the fp8 grid against the bf16 grid of the SAME latents.

Stated tolerance (full-width geo decoder, 513^3 grid): **the fp8 surface lies inside the 6e-2 level band of the bf16 field, and the
bf16 surface inside the fp8 field's** -- at least 99.9 % of the vertices of either mesh sit where the OTHER grid's trilinearly
interpolated |logit| is <= 6e-2 of that grid's largest |logit|.  This is what the arithmetic guarantees and what the test enforces.

How far that band is from the surface in SPACE depends on the field's gradient there, which is a property of the weights -- and the
measurement below says plainly that on a smooth synthetic field it is far: with the Fourier features above frequency 2 switched off
in `query_proj` (an object-like field, 0.9 M vertices at 513^3 instead of the noise-like 15 M) the fp8 mesh has 5x the vertices of the
bf16 mesh and 73 % of its vertices are more than one voxel from it -- the field is within 6e-2 of zero over wide regions, and the
incoherent rounding noise fragments the surface there.  An occupancy-logit field of a trained decoder is steep at the surface; whether
it is steep enough is one of the things `tools/verify_checkpoint.py`-day has to measure.  Until then **configs[3]'s fp8 mode is a
measured option (profiles/r06_bench_cfg4.json: 0.22 objects/s at 513^3), not an accepted default**, and the voxel distances are
reported by this test (Chebyshev voxel distance between the vertex sets by dilation of the occupancy), not bounded."""
import numpy as np
import pytest

from parity_support import TOL, bf16_round_matrices, report

pytestmark = pytest.mark.gpu

R = 512


def voxel_distances(va, vb, n, kmax=4):
    """for every vertex of `va` (index coordinates, [N, 3] on the GPU): Chebyshev distance, in voxels, from its voxel to the nearest
    voxel that holds a vertex of `vb`; kmax + 1 = farther than kmax.  Dilation of vb's occupancy by 3 x 3 x 3 max-pooling."""
    import torch
    import torch.nn.functional as F
    occ = torch.zeros(n * n * n, dtype=torch.float16, device=va.device)
    ib = vb.floor().long().clamp_(0, n - 1)
    occ[(ib[:, 0] * n + ib[:, 1]) * n + ib[:, 2]] = 1.0
    occ = occ.view(1, 1, n, n, n)
    ia = va.floor().long().clamp_(0, n - 1)
    flat = (ia[:, 0] * n + ia[:, 1]) * n + ia[:, 2]
    dist = torch.full((va.shape[0],), kmax + 1, dtype=torch.int32, device=va.device)
    for k in range(kmax + 1):
        hit = occ.view(-1)[flat] > 0
        dist = torch.where(hit & (dist > k), torch.full_like(dist, k), dist)
        if k < kmax:
            occ = F.max_pool3d(occ, 3, 1, 1)
    return dist


def interp(grid, v):
    """trilinear interpolation of grid [n, n, n] at index coordinates v [N, 3] (both on the GPU)"""
    import torch
    n = grid.shape[0]
    i0 = v.floor().long().clamp_(0, n - 2)
    f = (v - i0.to(v.dtype)).clamp_(0.0, 1.0)
    flat = grid.reshape(-1)
    out = torch.zeros(v.shape[0], dtype=torch.float32, device=v.device)
    for dx in (0, 1):
        for dy in (0, 1):
            for dz in (0, 1):
                w = (f[:, 0] if dx else 1 - f[:, 0]) * (f[:, 1] if dy else 1 - f[:, 1]) * (f[:, 2] if dz else 1 - f[:, 2])
                out += w * flat[((i0[:, 0] + dx) * n + (i0[:, 1] + dy)) * n + (i0[:, 2] + dz)]
    return out


def surfaces(sd, cfg, seed):
    """[(grid, vertices in index coordinates, faces)] for the bf16 and the fp8 mode of the geo decoder, one set of latents, 513^3"""
    import torch
    from r3g import ffi, mc as gpu_mc, model as M
    L = ffi.lib()
    gpu = M.ShapeModel(cfg, sd, 0)
    lat = torch.randn(cfg["vae"]["num_latents"], cfg["vae"]["embed_dim"], generator=torch.Generator().manual_seed(seed))
    gpu.vae_decode(lat)
    out = []
    try:
        for mode in (0, 1):
            ffi.check(L.r3g_set_option(b"geo_fp8", mode))
            grid = gpu.grid_query(1.01, R)
            v, f = gpu_mc.marching_cubes(grid, 0.0)
            out.append((grid, v.clone(), int(f.shape[0])))
    finally:
        ffi.check(L.r3g_set_option(b"geo_fp8", 0))
    del gpu
    torch.cuda.empty_cache()
    return out


@pytest.mark.parametrize("field", ["object-like", "noise-like"])
def test_fp8_geo_decoder_mesh_stays_on_the_bf16_mesh_at_513(field):
    import torch
    from oracle import hy3d_torch as H
    cfg = H.wide_config(depth=1, depth_single=1, vae_layers=1, cond_layers=1)
    sd = bf16_round_matrices(H.synthetic_state_dict(cfg, seed=11))
    if field == "object-like":
        # FourierEmbedder layout (oracle/hy3d_torch.py): [x y z | sin(e) 24 | cos(e) 24], e = coordinate-major x 8
        # frequencies 2^k: keep k <= 1
        w = sd["vae.geo_decoder.query_proj.weight"].clone()
        for c in range(3):
            for k in range(2, 8):
                w[:, 3 + c * 8 + k] = 0
                w[:, 27 + c * 8 + k] = 0
        sd["vae.geo_decoder.query_proj.weight"] = w
    (ga, va, fa), (gb, vb, fb) = surfaces(sd, cfg, 21)
    n = R + 1
    na, nb = int(va.shape[0]), int(vb.shape[0])
    assert na > 1000 and nb > 1000 and fa > 0 and fb > 0
    # ---- the stated tolerance: each surface inside the other field's 6e-2 level band
    eps = TOL["grid_logits_fp8"]
    band_b = interp(ga, vb).abs() / float(ga.abs().max())        # fp8 vertices in the bf16 field
    band_a = interp(gb, va).abs() / float(gb.abs().max())        # bf16 vertices in the fp8 field
    inside = min(float((band_b <= eps).float().mean()), float((band_a <= eps).float().mean()))
    report("configs[3] %s field, 513^3 (vertices bf16 %d / fp8 %d): vertices OUTSIDE the other field's 6e-2 level band" % (field, na, nb),
           1.0 - inside, 1e-3)
    report("configs[3] %s field: largest |other field| at a vertex, over its largest |logit|" % field,
           max(float(band_a.max()), float(band_b.max())), 2 * eps)
    assert inside >= 0.999
    assert max(float(band_a.max()), float(band_b.max())) <= 2 * eps
    del ga, gb, band_a, band_b
    torch.cuda.empty_cache()
    # ---- reported, not bounded: how far apart the two vertex sets are in space (a property of the field's gradient, see the docstring)
    d_ab = voxel_distances(va, vb, n)
    d_ba = voxel_distances(vb, va, n)
    near = min(float((d_ab <= 1).float().mean()), float((d_ba <= 1).float().mean()))
    far = max(float((d_ab > 3).float().mean()), float((d_ba > 3).float().mean()))
    chamfer = 0.5 * (float(d_ab.clamp(max=5).float().mean()) + float(d_ba.clamp(max=5).float().mean()))
    report("configs[3] %s field: vertices farther than ONE voxel from the other mesh (reported, not bounded)" % field, 1.0 - near, 1.0)
    report("configs[3] %s field: vertices farther than 3 voxels (reported)" % field, far, 1.0)
    report("configs[3] %s field: mean Chebyshev voxel distance between the vertex sets (reported)" % field, chamfer, 5.0)
    if field == "object-like":
        assert na < 3_000_000, "the low-frequency checkpoint should give a smooth surface, not a noise field: %d vertices" % na
