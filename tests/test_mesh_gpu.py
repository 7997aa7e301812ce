"""GPU parity of the mesh cleaners (include/r3g.h "mesh cleaners") against the numpy restatement
oracle/mesh_clean.py: faces AND float32 vertices bit-exact, on marching-cubes meshes with floaters, on triangle
soups with degenerate and duplicate faces, and on the empty / tiny edge cases."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mc_mesh(n=65, seed=0, floaters=3):
    """smooth blob + a few small far-away spheres, meshed by the product marching cubes"""
    from r3g import mc
    ax = torch.linspace(-1, 1, n)
    X, Y, Z = torch.meshgrid(ax, ax, ax, indexing="ij")
    rng = np.random.default_rng(seed)
    f = torch.full((n, n, n), -1.0)
    for _ in range(4):
        c = rng.uniform(-0.3, 0.3, 3)
        f = torch.maximum(f, float(rng.uniform(0.25, 0.45)) - torch.sqrt((X - c[0]) ** 2 + (Y - c[1]) ** 2 + (Z - c[2]) ** 2))
    for _ in range(floaters):
        c = rng.uniform(0.75, 0.9, 3) * rng.choice([-1, 1], 3)
        f = torch.maximum(f, 0.05 - torch.sqrt((X - c[0]) ** 2 + (Y - c[1]) ** 2 + (Z - c[2]) ** 2))
    v, fa = mc.extract_mesh(f.cuda().contiguous(), 0.0, 1.01, n - 1)
    return v, fa


def _soup(nv, nf, seed):
    rng = np.random.default_rng(seed)
    v = rng.standard_normal((nv, 3)).astype(np.float32)
    f = rng.integers(0, nv, (nf, 3)).astype(np.int32)
    f[::7, 1] = f[::7, 0]                      # degenerate faces
    f[5::11] = f[4::11][:len(f[5::11])]        # exact duplicates
    f[9::13] = f[8::13][:len(f[9::13]), ::-1]  # same vertex set, other winding
    return torch.from_numpy(v).cuda(), torch.from_numpy(f).cuda()


def _same(got, want):
    gv, gf = got[0].cpu().numpy(), got[1].cpu().numpy()
    wv, wf = want
    assert gf.shape == wf.shape and np.array_equal(gf, wf)
    assert gv.shape == wv.shape and np.array_equal(gv.view(np.uint32), wv.view(np.uint32))


@pytest.mark.parametrize("seed", [0, 1])
def test_floaters_and_degenerate_on_mc_mesh(seed):
    from oracle import mesh_clean
    from r3g import meshops
    v, f = _mc_mesh(65, seed)
    hv, hf = v.cpu().numpy(), f.cpu().numpy()
    got = meshops.remove_floaters(v, f, 0.02)
    want = mesh_clean.remove_floaters(hv, hf, 0.02)
    assert len(want[1]) < len(hf)                         # the small spheres went away
    _same(got, want)
    _same(meshops.remove_floaters(v, f), mesh_clean.remove_floaters(hv, hf))             # default ratio
    _same(meshops.remove_floaters(v, f, 0.0), mesh_clean.remove_floaters(hv, hf, 0.0))   # keeps everything
    _same(meshops.remove_degenerate(*got), mesh_clean.remove_degenerate(*want))


@pytest.mark.parametrize("nv,nf,seed", [(50, 400, 0), (5000, 20000, 1), (3, 1, 2)])
def test_soup(nv, nf, seed):
    from oracle import mesh_clean
    from r3g import meshops
    v, f = _soup(nv, nf, seed)
    hv, hf = v.cpu().numpy(), f.cpu().numpy()
    _same(meshops.remove_degenerate(v, f), mesh_clean.remove_degenerate(hv, hf))
    _same(meshops.remove_floaters(v, f, 0.3), mesh_clean.remove_floaters(hv, hf, 0.3))
    _same(meshops.cluster_faces(v, f, max(1, nf // 10)), mesh_clean.reduce_faces(hv, hf, max(1, nf // 10)))


def _touching_parts():
    """a big closed blob and a small octahedron that touches it in exactly ONE shared vertex (no shared edge), plus a fan of
    three faces around one non-manifold edge: the fixture on which shared-vertex and shared-edge connectivity differ"""
    v, f = _mc_mesh(33, 5, floaters=0)
    hv, hf = v.cpu().numpy(), f.cpu().numpy().astype(np.int64)
    n = len(hv)
    p = hv[7]                                                   # the shared vertex: index 7 of the blob
    d = 0.02
    extra = np.array([p + [d, d, 0], p + [d, -d, 0], p + [2 * d, 0, 0], p + [d, 0, d], p + [d, 0, -d]], np.float32)
    a, b, c, top, bot = n, n + 1, n + 2, n + 3, n + 4            # octahedron: apex 7 .. ring (a, top, b, bot) .. apex c
    octa = np.array([[7, a, top], [7, top, b], [7, b, bot], [7, bot, a], [c, top, a], [c, b, top], [c, bot, b], [c, a, bot]])
    m = n + 5                                                   # three faces around the non-manifold edge (m, m+1), far away
    fan_v = np.array([[5, 5, 5], [5, 5, 6], [6, 5, 5], [5, 6, 5], [4, 4, 5]], np.float32)
    fan = np.array([[m, m + 1, m + 2], [m, m + 1, m + 3], [m, m + 1, m + 4]])
    return np.concatenate([hv, extra, fan_v]), np.concatenate([hf, octa, fan]).astype(np.int32), len(hf)


def test_floaters_are_joined_through_edges_not_vertices():
    """MeshLab's small-component filter walks face-face adjacency: a part hanging on the surface by a single vertex is a
    component of its own (and goes away when it is small); rounds 1-2 joined it to the surface through the vertex."""
    from oracle import mesh_clean
    from r3g import ffi, meshops
    hv, hf, n_blob = _touching_parts()
    v, f = torch.from_numpy(hv).cuda(), torch.from_numpy(hf).cuda()
    lab_e = mesh_clean.face_components(hf, len(hv))
    lab_v = mesh_clean.face_components(hf, len(hv), by_vertex=True)
    assert len(set(lab_e)) == 3 and len(set(lab_v)) == 2         # blob | octahedron | fan   vs   blob+octahedron | fan
    assert len(set(lab_e[-3:])) == 1                              # the three faces around the non-manifold edge: one part
    got = meshops.remove_floaters(v, f, 0.05)
    want = mesh_clean.remove_floaters(hv, hf, 0.05)
    _same(got, want)
    assert len(want[1]) == n_blob                                 # octahedron and fan removed, the blob untouched
    L = ffi.lib()
    try:
        ffi.check(L.r3g_set_option(b"floater_by_vertex", 1))
        got_v = meshops.remove_floaters(v, f, 0.05)
    finally:
        ffi.check(L.r3g_set_option(b"floater_by_vertex", 0))
    want_v = mesh_clean.remove_floaters(hv, hf, 0.05, by_vertex=True)
    _same(got_v, want_v)
    assert len(want_v[1]) == n_blob + 8                           # the octahedron survives when vertices join components


@pytest.mark.parametrize("n,budget", [(65, 3000), (129, 40000), (129, 500)])
def test_cluster_on_mc_mesh(n, budget):
    from oracle import mesh_clean
    from r3g import meshops
    v, f = _mc_mesh(n, 3, floaters=0)
    want = mesh_clean.reduce_faces(v.cpu().numpy(), f.cpu().numpy(), budget)
    got = meshops.cluster_faces(v, f, budget)
    assert 0 < len(want[1]) <= budget
    _same(got, want)
    again = meshops.cluster_faces(*got, budget)           # already within budget: untouched
    _same(again, want)


def test_inputs_untouched_and_empty():
    from r3g import meshops
    v, f = _soup(100, 300, 5)
    v0, f0 = v.clone(), f.clone()
    meshops.remove_degenerate(v, f)
    assert torch.equal(v, v0) and torch.equal(f, f0)
    ev, ef = meshops.remove_floaters(torch.zeros(0, 3, device="cuda"), torch.zeros(0, 3, dtype=torch.int32, device="cuda"))
    assert ev.shape == (0, 3) and ef.shape == (0, 3)
    with pytest.raises(ValueError):
        meshops.remove_degenerate(v.cpu(), f.cpu())       # no CPU path


def test_postprocessor_classes_keep_the_mesh_on_the_gpu():
    import emu_qem
    from hy3dgen.shapegen import DegenerateFaceRemover, FaceReducer, FloaterRemover
    from oracle import mesh_clean
    from r3g.mesh import Mesh
    v, f = _mc_mesh(65, 4)
    m = Mesh.from_device(v, f)
    for c in (FloaterRemover(), DegenerateFaceRemover(), FaceReducer()):
        m = c(m)
    assert m._v is None and m.n_faces <= 40000            # nothing was downloaded on the way
    w = mesh_clean.remove_floaters(v.cpu().numpy(), f.cpu().numpy())
    w = mesh_clean.remove_degenerate(*w)
    w = emu_qem.reduce_faces(*w, 40000)                   # FaceReducer = quadric edge collapse (tests/emu/qem_emu.cpp)
    assert np.array_equal(m.faces, w[1]) and np.array_equal(m.vertices.astype(np.float32), w[0])
    host = Mesh(v.cpu().numpy(), f.cpu().numpy())         # host-born mesh: uploaded, same answer
    h = FaceReducer()(host, max_facenum=2000)
    w2 = emu_qem.reduce_faces(v.cpu().numpy(), f.cpu().numpy(), 2000)
    assert np.array_equal(h.faces, w2[1])
    data = h.export(file_type="glb")
    assert data[:4] == b"glTF"


def test_full_size_properties():
    """At the size of a dense 257^3 extraction (millions of vertices) the cleaners are checked through size-independent
    properties: dense vertex indexing, idempotence, the face budget, and agreement of the survivors with the input."""
    from r3g import mc, meshops
    g = torch.Generator().manual_seed(3)
    lo = torch.randn(1, 1, 33, 33, 33, generator=g)
    vol = torch.nn.functional.interpolate(lo, size=(257, 257, 257), mode="trilinear", align_corners=True)[0, 0]
    v, f = mc.extract_mesh(vol.cuda().contiguous(), 0.0, 1.01, 256)
    assert v.shape[0] > 1000000
    for fn, args in ((meshops.remove_floaters, (0.005,)), (meshops.remove_degenerate, ()), (meshops.reduce_faces, (40000,))):
        a, b = fn(v, f, *args)
        assert 0 < b.shape[0] <= f.shape[0] and a.shape[0] <= v.shape[0]
        u = torch.unique(b)
        assert u.numel() == a.shape[0] and int(u[0]) == 0 and int(u[-1]) == a.shape[0] - 1
        a2, b2 = fn(a, b, *args)
        assert torch.equal(a2, a) and torch.equal(b2, b)
    fv, ff = meshops.remove_floaters(v, f, 0.005)
    # survivors are input faces, in input order, with their vertices' coordinates unchanged
    tri_in = v[f.long()].reshape(-1, 9)
    tri_out = fv[ff.long()].reshape(-1, 9)
    idx = torch.arange(0, tri_out.shape[0], max(1, tri_out.shape[0] // 4096), device="cuda")
    first = tri_out[idx]
    # every sampled output triangle occurs in the input (compared through a position-weighted checksum)
    h_in = (tri_in * torch.arange(1, 10, device="cuda")).sum(1)
    h_out = (first * torch.arange(1, 10, device="cuda")).sum(1)
    assert torch.isin(h_out, h_in).all()
    rv, rf = meshops.reduce_faces(fv, ff, 40000)
    assert rf.shape[0] <= 40000
    lo_b, hi_b = fv.min(0).values, fv.max(0).values
    ext = (hi_b - lo_b).max()
    assert (rv >= lo_b - 0.01 * ext).all() and (rv <= hi_b + 0.01 * ext).all()   # optimal placements stay at the surface


def test_results_do_not_depend_on_scheduling():
    """union-find hooking, the dedup table and the integer cluster sums race by design; the results must not: repeated
    runs on a 100 k-vertex mesh and on a soup full of duplicates are identical bit for bit"""
    from r3g import meshops
    v, f = _mc_mesh(129, 9, floaters=4)
    sv, sf = _soup(20000, 120000, 4)
    ref = None
    for _ in range(6):
        a = meshops.remove_floaters(v, f, 0.01)
        b = meshops.reduce_faces(*a, 5000)                 # quadric edge collapse on the marching-cubes mesh
        c = meshops.cluster_faces(sv, sf, 3000)            # vertex clustering on the soup
        d = meshops.remove_floaters(sv, sf, 0.5)
        cur = [t.clone() for pair in (a, b, c, d) for t in pair]
        if ref is None:
            ref = cur
        else:
            assert all(torch.equal(x, y) for x, y in zip(ref, cur))
