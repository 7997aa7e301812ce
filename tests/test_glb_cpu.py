"""The GLB writer against an independent glTF 2.0 validator (tests/gltf_validate.py, written from the specification and
sharing no code with r3g/mesh.py): what a third-party loader decodes must equal what went in."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "3d-re-gen_amd"))
sys.path.insert(0, HERE)

from gltf_validate import validate_glb  # noqa: E402
from r3g.mesh import Mesh, encode_png  # noqa: E402


def _mesh(nv=37, nf=55, seed=0):
    rng = np.random.default_rng(seed)
    v = rng.normal(size=(nv, 3))
    f = np.stack([rng.permutation(nv)[:3] for _ in range(nf)])
    return v, f


@pytest.mark.parametrize("nv,nf", [(3, 1), (37, 55), (1001, 1999)])
def test_plain_mesh(nv, nf):
    v, f = _mesh(nv, nf, nv)
    got = validate_glb(Mesh(v, f).to_glb())
    assert np.array_equal(got["positions"], v.astype(np.float32))
    assert np.array_equal(got["indices"].astype(np.int64), f)
    assert got["image"] is None and set(got["attributes"]) == {"POSITION"}


def test_vertex_colours_rgb_and_rgba():
    v, f = _mesh()
    rng = np.random.default_rng(5)
    for c in (3, 4):
        col = rng.integers(0, 256, (len(v), c), dtype=np.uint8)
        got = validate_glb(Mesh(v, f, vertex_colors=col).to_glb())
        assert np.array_equal(got["attributes"]["COLOR_0"][:, :c], col)
        if c == 3:
            assert (got["attributes"]["COLOR_0"][:, 3] == 255).all()


@pytest.mark.parametrize("channels,shape", [(3, (5, 7)), (4, (16, 16)), (3, (33, 2))])
def test_base_colour_texture(channels, shape):
    v, f = _mesh(50, 80, 9)
    rng = np.random.default_rng(11)
    uv = rng.random((len(v), 2), dtype=np.float32)
    tex = rng.integers(0, 256, shape + (channels,), dtype=np.uint8)
    got = validate_glb(Mesh(v, f, uv=uv, texture=tex).to_glb())
    assert np.array_equal(got["attributes"]["TEXCOORD_0"], uv)
    assert np.array_equal(got["image"], tex)
    doc = got["doc"]
    assert doc["meshes"][0]["primitives"][0]["material"] == 0
    assert doc["materials"][0]["pbrMetallicRoughness"]["baseColorTexture"]["index"] == 0


def test_png_encoder_is_read_by_pillow():
    Image = pytest.importorskip("PIL.Image")
    import io
    rng = np.random.default_rng(3)
    for c in (3, 4):
        tex = rng.integers(0, 256, (9, 13, c), dtype=np.uint8)
        back = np.asarray(Image.open(io.BytesIO(encode_png(tex))))
        assert np.array_equal(back, tex)


def test_uv_follows_vertex_compaction():
    v, f = _mesh(20, 6, 2)
    uv = np.arange(40, dtype=np.float32).reshape(20, 2)
    m = Mesh(v, f, uv=uv, texture=np.zeros((2, 2, 3), np.uint8))
    m.remove_unreferenced_vertices()
    used = np.unique(f.reshape(-1))
    assert np.array_equal(m.uv, uv[used])
    validate_glb(m.to_glb())


def test_validator_rejects_broken_files():
    v, f = _mesh()
    good = Mesh(v, f).to_glb()
    validate_glb(good)
    with pytest.raises(AssertionError):
        validate_glb(good[:-4])                                 # truncated: header length no longer matches
    bad = bytearray(good)
    # first float of the POSITION data: the accessor's min/max no longer bound the data
    import json
    import struct
    jlen = struct.unpack_from("<I", good, 12)[0]
    doc = json.loads(good[20:20 + jlen])
    off = 20 + jlen + 8 + doc["bufferViews"][0]["byteOffset"]
    struct.pack_into("<f", bad, off, 1.0e9)
    with pytest.raises(AssertionError):
        validate_glb(bytes(bad))
    bad = bytearray(good)
    ioff = 20 + jlen + 8 + doc["bufferViews"][1]["byteOffset"]
    struct.pack_into("<I", bad, ioff, 10 ** 6)                  # index past the vertex count
    with pytest.raises(AssertionError):
        validate_glb(bytes(bad))
