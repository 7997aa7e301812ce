"""Minimal Trimesh-compatible mesh + binary glTF (GLB) writer.

`trimesh` is what the reference stage holds between shapegen, the cleaners and `mesh.export(path)`
(src/2d_to_3d_models/run.py:84-102); it is not installed in this image, so this class provides the
attributes that script touches: .vertices, .faces, .is_empty, .export(path), update_vertices,
update_faces, remove_unreferenced_vertices, nondegenerate_faces, process.  The GLB carries POSITION
(float32) + uint32 indices (+ optional per-vertex COLOR_0, + optional TEXCOORD_0 with a PNG baseColorTexture), which is
what the downstream consumer loads with load_textures=True (src/scene_reconstruction/source/pose_matching_planar.py:882-906).
"""
import json
import struct
import zlib

import numpy as np


class Mesh:
    """Host arrays (float64 vertices, int64 faces, as trimesh holds them) and/or device buffers (float32 / int32
    torch CUDA tensors, as marching cubes and the GPU cleaners produce them).  The host arrays of a device-born mesh
    are only downloaded when something reads `.vertices` / `.faces`; assigning either drops the device copy."""

    def __init__(self, vertices=None, faces=None, vertex_colors=None, process=True, uv=None, texture=None):
        self._dv = self._df = None
        self.uv = None if uv is None else np.asarray(uv, np.float32).reshape(-1, 2)        # glTF convention: v = 0 at the top
        self.texture = None if texture is None else np.asarray(texture, np.uint8)          # [H, W, 3 | 4]
        self._v = np.zeros((0, 3), np.float64) if vertices is None else np.asarray(vertices, np.float64).reshape(-1, 3)
        self._f = np.zeros((0, 3), np.int64) if faces is None else np.asarray(faces, np.int64).reshape(-1, 3)
        self.vertex_colors = None if vertex_colors is None else np.asarray(vertex_colors, np.uint8)
        self.metadata = {}

    @classmethod
    def from_device(cls, verts, faces, metadata=None):
        """verts float32 [V,3], faces int32 [F,3] torch CUDA tensors (kept, not copied)"""
        m = cls.__new__(cls)
        m._v = m._f = None
        m._dv, m._df = verts, faces
        m.vertex_colors = None
        m.uv = m.texture = None
        m.metadata = dict(metadata or {})
        return m

    # -- storage ---------------------------------------------------------------------------------
    @property
    def vertices(self):
        if self._v is None:
            self._v = self._dv.detach().cpu().numpy().astype(np.float64).reshape(-1, 3)
        return self._v

    @vertices.setter
    def vertices(self, value):
        self._host()
        self._v = np.asarray(value, np.float64).reshape(-1, 3)
        self._dv = self._df = None

    @property
    def faces(self):
        if self._f is None:
            self._f = self._df.detach().cpu().numpy().astype(np.int64).reshape(-1, 3)
        return self._f

    @faces.setter
    def faces(self, value):
        self._host()
        self._f = np.asarray(value, np.int64).reshape(-1, 3)
        self._dv = self._df = None

    def _host(self):
        return self.vertices, self.faces

    @property
    def n_vertices(self):
        return int(self._v.shape[0] if self._v is not None else self._dv.shape[0])

    @property
    def n_faces(self):
        return int(self._f.shape[0] if self._f is not None else self._df.shape[0])

    def device_buffers(self, device=None):
        """(verts float32 [V,3], faces int32 [F,3]) on the GPU; uploads the host arrays when there is no device copy"""
        if self._dv is None:
            import torch
            if not torch.cuda.is_available():
                raise RuntimeError("Mesh.device_buffers: no GPU (the mesh cleaners have no CPU path)")
            dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
            self._dv = torch.from_numpy(np.ascontiguousarray(self._v, np.float32)).to(dev)
            self._df = torch.from_numpy(np.ascontiguousarray(self._f, np.int32)).to(dev)
        return self._dv, self._df

    # -- trimesh-like surface ------------------------------------------------------------------
    @property
    def is_empty(self):
        return self.n_vertices == 0 or self.n_faces == 0

    def copy(self):
        if self._v is None:
            m = Mesh.from_device(self._dv.clone(), self._df.clone(), self.metadata)
        else:
            m = Mesh(self.vertices.copy(), self.faces.copy())
            m.metadata = dict(self.metadata)
        m.vertex_colors = None if self.vertex_colors is None else self.vertex_colors.copy()
        m.uv = None if getattr(self, "uv", None) is None else self.uv.copy()
        m.texture = getattr(self, "texture", None)
        return m

    def update_vertices(self, mask):
        mask = np.asarray(mask, bool)
        remap = np.cumsum(mask) - 1
        keep_face = mask[self.faces].all(axis=1)
        self.faces = remap[self.faces[keep_face]]
        self.vertices = self.vertices[mask]
        if self.vertex_colors is not None:
            self.vertex_colors = self.vertex_colors[mask]
        if getattr(self, "uv", None) is not None:
            self.uv = self.uv[mask]

    def update_faces(self, mask):
        self.faces = self.faces[np.asarray(mask)]

    def nondegenerate_faces(self):
        f = self.faces
        return (f[:, 0] != f[:, 1]) & (f[:, 1] != f[:, 2]) & (f[:, 0] != f[:, 2])

    def remove_unreferenced_vertices(self):
        used = np.zeros(len(self.vertices), bool)
        used[self.faces.reshape(-1)] = True
        self.update_vertices(used)

    def simplify_quadric_decimation(self, face_count=None, percent=None, **kwargs):
        """trimesh.Trimesh.simplify_quadric_decimation as the reference's remesh step calls it (run.py:47-49): a new mesh
        with <= face_count faces by quadric edge collapse -- the GPU decimator of r3g.meshops (there is no CPU path)"""
        from . import meshops
        if face_count is None:
            if percent is None:
                raise ValueError("face_count or percent is required")
            face_count = int(self.n_faces * (1.0 - float(percent)))
        if self.is_empty or self.n_faces <= face_count:
            return self.copy()
        v, f = meshops.reduce_faces(*self.device_buffers(), int(face_count))
        return Mesh.from_device(v, f, self.metadata)

    def process(self, validate=False):
        """merge bit-identical vertices (trimesh.Trimesh.process default), drop degenerate faces if validate"""
        if len(self.vertices):
            _, first, inv = np.unique(self.vertices, axis=0, return_index=True, return_inverse=True)
            order = np.argsort(first)
            rank = np.empty_like(order)
            rank[order] = np.arange(len(order))
            self.faces = rank[inv.reshape(-1)][self.faces]
            self.vertices = self.vertices[first[order]]
            if self.vertex_colors is not None:
                self.vertex_colors = self.vertex_colors[first[order]]
            if getattr(self, "uv", None) is not None:     # (vertices that differ only in uv are merged: first one wins)
                self.uv = self.uv[first[order]]
        if validate:
            self.update_faces(self.nondegenerate_faces())
        return self

    # -- GLB ---------------------------------------------------------------------------------------
    def to_glb(self):
        v = np.ascontiguousarray(self.vertices, np.float32)
        f = np.ascontiguousarray(self.faces, np.uint32).reshape(-1)
        chunks, views, accessors = [], [], []

        def add(data, target):
            off = sum(len(c) for c in chunks)
            raw = data.tobytes()
            raw += b"\x00" * (-len(raw) % 4)
            chunks.append(raw)
            views.append({"buffer": 0, "byteOffset": off, "byteLength": data.nbytes})
            if target is not None:
                views[-1]["target"] = target
            return len(views) - 1

        accessors.append({"bufferView": add(v, 34962), "componentType": 5126, "count": int(len(v)), "type": "VEC3",
                          "min": v.min(axis=0).tolist() if len(v) else [0, 0, 0],
                          "max": v.max(axis=0).tolist() if len(v) else [0, 0, 0]})
        accessors.append({"bufferView": add(f, 34963), "componentType": 5125, "count": int(len(f)), "type": "SCALAR"})
        attrs = {"POSITION": 0}
        if self.vertex_colors is not None and len(self.vertex_colors) == len(v):
            c = np.ascontiguousarray(self.vertex_colors[:, :4] if self.vertex_colors.shape[1] >= 4 else
                                     np.concatenate([self.vertex_colors, np.full((len(v), 1), 255, np.uint8)], 1), np.uint8)
            accessors.append({"bufferView": add(c, 34962), "componentType": 5121, "count": int(len(v)), "type": "VEC4",
                              "normalized": True})
            attrs["COLOR_0"] = 2
        extra = {}
        prim = {"attributes": attrs, "indices": 1, "mode": 4}
        uv = getattr(self, "uv", None)
        tex = getattr(self, "texture", None)
        if uv is not None and tex is not None and len(uv) == len(v):
            t = np.ascontiguousarray(uv, np.float32)
            accessors.append({"bufferView": add(t, 34962), "componentType": 5126, "count": int(len(t)), "type": "VEC2"})
            attrs["TEXCOORD_0"] = len(accessors) - 1
            png = np.frombuffer(encode_png(tex), np.uint8)
            img_view = add(png, None)
            extra = {"images": [{"bufferView": img_view, "mimeType": "image/png"}],
                     "samplers": [{"magFilter": 9729, "minFilter": 9729, "wrapS": 33071, "wrapT": 33071}],
                     "textures": [{"sampler": 0, "source": 0}],
                     "materials": [{"pbrMetallicRoughness": {"baseColorTexture": {"index": 0}, "metallicFactor": 0.0,
                                                             "roughnessFactor": 1.0}}]}
            prim["material"] = 0
        bin_blob = b"".join(chunks)
        doc = {"asset": {"version": "2.0", "generator": "r3g"}, "scene": 0, "scenes": [{"nodes": [0]}],
               "nodes": [{"mesh": 0}], "meshes": [{"primitives": [prim]}],
               "buffers": [{"byteLength": len(bin_blob)}], "bufferViews": views, "accessors": accessors}
        doc.update(extra)
        js = json.dumps(doc, separators=(",", ":")).encode()
        js += b" " * (-len(js) % 4)
        total = 12 + 8 + len(js) + 8 + len(bin_blob)
        return b"".join([struct.pack("<4sII", b"glTF", 2, total), struct.pack("<I4s", len(js), b"JSON"), js,
                         struct.pack("<I4s", len(bin_blob), b"BIN\x00"), bin_blob])

    def export(self, path=None, file_type=None):
        path_s = None if path is None else str(path)
        kind = (file_type or (path_s.rsplit(".", 1)[-1] if path_s else "glb")).lower()
        if kind == "glb":
            data = self.to_glb()
        elif kind == "obj":
            lines = ["v %.9g %.9g %.9g" % tuple(p) for p in self.vertices]
            lines += ["f %d %d %d" % tuple(t + 1) for t in self.faces]
            data = ("\n".join(lines) + "\n").encode()
        else:
            raise ValueError("unsupported export type: " + kind)
        if path_s is not None:
            with open(path_s, "wb") as fh:
                fh.write(data)
        return data


def encode_png(img):
    """uint8 [H, W, 3 | 4] -> PNG bytes (8-bit truecolour, filter 0 on every scanline)"""
    img = np.ascontiguousarray(img, np.uint8)
    if img.ndim != 3 or img.shape[2] not in (3, 4):
        raise ValueError("texture must be [H, W, 3] or [H, W, 4] uint8")
    h, w, c = img.shape

    def chunk(tag, body):
        return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xFFFFFFFF)
    raw = np.concatenate([np.zeros((h, 1), np.uint8), img.reshape(h, w * c)], axis=1).tobytes()
    return b"".join([b"\x89PNG\r\n\x1a\n", chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2 if c == 3 else 6, 0, 0, 0)),
                     chunk(b"IDAT", zlib.compress(raw, 6 if h * w <= 512 * 512 else 1)), chunk(b"IEND", b"")])


def load_glb(data):
    """Parse a GLB written by Mesh.to_glb (tests / round trips)."""
    if isinstance(data, str):
        with open(data, "rb") as fh:
            data = fh.read()
    magic, ver, total = struct.unpack_from("<4sII", data, 0)
    assert magic == b"glTF" and ver == 2 and total == len(data)
    jl, jt = struct.unpack_from("<I4s", data, 12)
    doc = json.loads(data[20:20 + jl])
    bl, bt = struct.unpack_from("<I4s", data, 20 + jl)
    blob = data[28 + jl:28 + jl + bl]
    prim = doc["meshes"][0]["primitives"][0]

    def read(acc_i, dtype, comps):
        a = doc["accessors"][acc_i]
        bv = doc["bufferViews"][a["bufferView"]]
        return np.frombuffer(blob, dtype, a["count"] * comps, bv["byteOffset"]).reshape(a["count"], comps) \
            if comps > 1 else np.frombuffer(blob, dtype, a["count"], bv["byteOffset"])
    v = read(prim["attributes"]["POSITION"], np.float32, 3)
    f = read(prim["indices"], np.uint32, 1).reshape(-1, 3)
    return Mesh(v, f)
