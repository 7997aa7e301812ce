"""An independent structural validator for binary glTF 2.0 (test infrastructure).

Written from the Khronos glTF 2.0 specification (sections 3.6 binary data storage, 3.7 geometry, 3.8/3.9 texture data and
materials, 4.4 GLB layout), NOT from r3g/mesh.py: it shares no code with the writer or with `load_glb`, parses the
container and every accessor itself and returns the decoded arrays, so that a test can compare what a third-party loader
would see with what went in.  Raises AssertionError with the violated rule."""
import json
import struct
import zlib

import numpy as np

COMPONENT = {5120: ("i1", 1), 5121: ("u1", 1), 5122: ("<i2", 2), 5123: ("<u2", 2), 5125: ("<u4", 4), 5126: ("<f4", 4)}
NCOMP = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT2": 4, "MAT3": 9, "MAT4": 16}


def _png_info(b):
    assert b[:8] == b"\x89PNG\r\n\x1a\n", "image: not a PNG signature"
    pos, idat, info, seen_end = 8, [], None, False
    while pos < len(b):
        (n,), tag = struct.unpack_from(">I", b, pos), b[pos + 4:pos + 8]
        body = b[pos + 8:pos + 8 + n]
        (crc,) = struct.unpack_from(">I", b, pos + 8 + n)
        assert crc == (zlib.crc32(tag + body) & 0xFFFFFFFF), "PNG chunk %r: bad CRC" % tag
        if tag == b"IHDR":
            info = struct.unpack(">IIBBBBB", body)
        elif tag == b"IDAT":
            idat.append(body)
        elif tag == b"IEND":
            seen_end = True
        pos += 12 + n
    assert info is not None and seen_end and idat, "PNG: IHDR / IDAT / IEND missing"
    w, h, depth, ctype, comp, flt, interlace = info
    assert depth == 8 and ctype in (2, 6) and comp == 0 and flt == 0 and interlace == 0, "PNG: unsupported header %r" % (info,)
    c = 3 if ctype == 2 else 4
    raw = zlib.decompress(b"".join(idat))
    assert len(raw) == h * (1 + w * c), "PNG: decompressed size does not match IHDR"
    rows = np.frombuffer(raw, np.uint8).reshape(h, 1 + w * c)
    prev = np.zeros(w * c, np.uint8)
    out = np.zeros((h, w * c), np.uint8)
    for y in range(h):   # undo the scanline filters (types 0-2 are enough for test images; 3/4 are refused)
        f, line = int(rows[y, 0]), rows[y, 1:].astype(np.int32)
        if f == 0:
            cur = line
        elif f == 1:
            cur = line.copy()
            for x in range(c, w * c):
                cur[x] = (cur[x] + cur[x - c]) & 255
        elif f == 2:
            cur = (line + prev) & 255
        else:
            raise AssertionError("PNG: filter type %d not handled by the validator" % f)
        out[y] = cur
        prev = out[y].astype(np.int32)
    return out.reshape(h, w, c)


def validate_glb(data):
    """returns {"positions", "indices", "attributes": {name: array}, "image": array|None, "doc": json}"""
    assert len(data) >= 20, "GLB shorter than header + one chunk header"
    magic, version, total = struct.unpack_from("<4sII", data, 0)
    assert magic == b"glTF", "magic"
    assert version == 2, "container version"
    assert total == len(data), "header length %d != file length %d" % (total, len(data))
    jlen, jtype = struct.unpack_from("<I4s", data, 12)
    assert jtype == b"JSON", "first chunk must be JSON"
    assert jlen % 4 == 0, "JSON chunk length not a multiple of 4"
    jraw = data[20:20 + jlen]
    assert jraw.rstrip(b" ") == jraw.rstrip(), "JSON chunk must be padded with spaces (0x20)"
    doc = json.loads(jraw.decode("utf-8"))
    pos = 20 + jlen
    blob = b""
    if pos < len(data):
        assert pos % 4 == 0, "BIN chunk not 4-byte aligned"
        blen, btype = struct.unpack_from("<I4s", data, pos)
        assert btype == b"BIN\x00", "second chunk must be BIN"
        assert blen % 4 == 0, "BIN chunk length not a multiple of 4"
        assert pos + 8 + blen == len(data), "chunks do not add up to the file length"
        blob = data[pos + 8:pos + 8 + blen]
    assert doc.get("asset", {}).get("version") == "2.0", "asset.version"
    bufs = doc.get("buffers", [])
    assert len(bufs) == 1 and "uri" not in bufs[0], "a GLB-stored buffer must be buffers[0] without uri"
    bl = bufs[0]["byteLength"]
    assert bl >= 1 and bl <= len(blob) and len(blob) - bl <= 3, "buffer.byteLength %d vs BIN chunk %d" % (bl, len(blob))
    views = doc.get("bufferViews", [])
    for i, bv in enumerate(views):
        assert bv["buffer"] == 0, "bufferView %d: buffer index" % i
        assert bv["byteLength"] >= 1, "bufferView %d: byteLength must be >= 1" % i
        assert bv.get("byteOffset", 0) + bv["byteLength"] <= bl, "bufferView %d runs past the buffer" % i
        if "target" in bv:
            assert bv["target"] in (34962, 34963), "bufferView %d: target" % i
        if "byteStride" in bv:
            assert 4 <= bv["byteStride"] <= 252 and bv["byteStride"] % 4 == 0, "bufferView %d: byteStride" % i

    def read_accessor(i):
        a = doc["accessors"][i]
        assert a["componentType"] in COMPONENT, "accessor %d: componentType" % i
        assert a["type"] in NCOMP, "accessor %d: type" % i
        assert a["count"] >= 1, "accessor %d: count must be >= 1" % i
        dt, size = COMPONENT[a["componentType"]]
        nc = NCOMP[a["type"]]
        bv = views[a["bufferView"]]
        off = bv.get("byteOffset", 0) + a.get("byteOffset", 0)
        assert a.get("byteOffset", 0) % size == 0 and off % size == 0, "accessor %d: not aligned to its component size" % i
        stride = bv.get("byteStride", size * nc)
        assert a.get("byteOffset", 0) + stride * (a["count"] - 1) + size * nc <= bv["byteLength"], "accessor %d does not fit its bufferView" % i
        if stride == size * nc:
            arr = np.frombuffer(blob, dt, a["count"] * nc, off).reshape(a["count"], nc)
        else:
            arr = np.stack([np.frombuffer(blob, dt, nc, off + k * stride) for k in range(a["count"])])
        if "min" in a or "max" in a:
            assert len(a["min"]) == nc and len(a["max"]) == nc, "accessor %d: min/max length" % i
            assert np.array_equal(np.asarray(a["min"], arr.dtype), arr.min(axis=0)), "accessor %d: min is not the data minimum" % i
            assert np.array_equal(np.asarray(a["max"], arr.dtype), arr.max(axis=0)), "accessor %d: max is not the data maximum" % i
        return a, bv, arr

    assert doc["scenes"] and 0 <= doc.get("scene", 0) < len(doc["scenes"]), "scene index"
    for n in doc["scenes"][doc.get("scene", 0)]["nodes"]:
        assert 0 <= n < len(doc["nodes"]), "scene node index"
    mesh_nodes = [n for n in doc["nodes"] if "mesh" in n]
    assert mesh_nodes, "no node references a mesh"
    for n in mesh_nodes:
        assert 0 <= n["mesh"] < len(doc["meshes"]), "node.mesh index"
    prim = doc["meshes"][mesh_nodes[0]["mesh"]]["primitives"][0]
    assert prim.get("mode", 4) == 4, "triangles expected"
    assert "POSITION" in prim["attributes"], "POSITION missing"
    out = {"attributes": {}, "image": None, "doc": doc}
    count = None
    for name, ai in prim["attributes"].items():
        a, bv, arr = read_accessor(ai)
        assert bv.get("target", 34962) == 34962, "%s: vertex data in an ELEMENT_ARRAY_BUFFER view" % name
        assert (bv.get("byteOffset", 0) + a.get("byteOffset", 0)) % 4 == 0, "%s: vertex attribute not 4-byte aligned" % name
        count = a["count"] if count is None else count
        assert a["count"] == count, "%s: attribute count differs from POSITION" % name
        if name == "POSITION":
            assert a["componentType"] == 5126 and a["type"] == "VEC3", "POSITION must be float VEC3"
            assert "min" in a and "max" in a, "POSITION accessor must carry min and max"
            assert np.isfinite(arr).all(), "POSITION holds NaN / inf"
        elif name == "TEXCOORD_0":
            assert a["type"] == "VEC2" and (a["componentType"] == 5126 or a.get("normalized")), "TEXCOORD_0 type"
        elif name == "COLOR_0":
            assert a["type"] in ("VEC3", "VEC4") and (a["componentType"] == 5126 or a.get("normalized")), "COLOR_0 type"
        out["attributes"][name] = arr
    out["positions"] = out["attributes"]["POSITION"]
    if "indices" in prim:
        a, bv, arr = read_accessor(prim["indices"])
        assert a["type"] == "SCALAR" and a["componentType"] in (5121, 5123, 5125), "indices type"
        assert bv.get("target", 34963) == 34963 and "byteStride" not in bv, "indices bufferView"
        assert a["count"] % 3 == 0, "index count not a multiple of 3"
        idx = arr.reshape(-1)
        assert int(idx.max()) < count, "index out of range"
        assert int(idx.max()) != {5121: 0xFF, 5123: 0xFFFF, 5125: 0xFFFFFFFF}[a["componentType"]], "primitive-restart value used"
        out["indices"] = idx.reshape(-1, 3)
    if "material" in prim:
        mat = doc["materials"][prim["material"]]
        bct = mat.get("pbrMetallicRoughness", {}).get("baseColorTexture")
        if bct is not None:
            assert bct.get("texCoord", 0) == 0 and "TEXCOORD_0" in prim["attributes"], "baseColorTexture needs TEXCOORD_0"
            tex = doc["textures"][bct["index"]]
            if "sampler" in tex:
                smp = doc["samplers"][tex["sampler"]]
                assert smp.get("wrapS", 10497) in (33071, 33648, 10497) and smp.get("wrapT", 10497) in (33071, 33648, 10497), "sampler wrap"
                assert smp.get("magFilter", 9729) in (9728, 9729), "sampler magFilter"
            img = doc["images"][tex["source"]]
            assert ("bufferView" in img) != ("uri" in img), "image needs exactly one of uri / bufferView"
            assert img["mimeType"] in ("image/png", "image/jpeg"), "image.mimeType"
            bv = views[img["bufferView"]]
            assert "target" not in bv and "byteStride" not in bv, "image bufferView must not have target / byteStride"
            raw = blob[bv.get("byteOffset", 0):bv.get("byteOffset", 0) + bv["byteLength"]]
            out["image"] = _png_info(raw)
    return out
