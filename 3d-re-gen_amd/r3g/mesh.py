"""Minimal Trimesh-compatible mesh + binary glTF (GLB) writer.

`trimesh` is what the reference stage holds between shapegen, the cleaners and `mesh.export(path)`
(src/2d_to_3d_models/run.py:84-102); it is not installed in this image, so this class provides the
attributes that script touches: .vertices, .faces, .is_empty, .export(path), update_vertices,
update_faces, remove_unreferenced_vertices, nondegenerate_faces, process.  The GLB carries POSITION
(float32) + uint32 indices (+ optional per-vertex COLOR_0), which is what the downstream consumer
loads (src/scene_reconstruction/source/pose_matching_planar.py:882-906).
"""
import json
import struct

import numpy as np


class Mesh:
    def __init__(self, vertices=None, faces=None, vertex_colors=None, process=True):
        self.vertices = np.zeros((0, 3), np.float64) if vertices is None else np.asarray(vertices, np.float64).reshape(-1, 3)
        self.faces = np.zeros((0, 3), np.int64) if faces is None else np.asarray(faces, np.int64).reshape(-1, 3)
        self.vertex_colors = None if vertex_colors is None else np.asarray(vertex_colors, np.uint8)
        self.metadata = {}

    # -- trimesh-like surface ------------------------------------------------------------------
    @property
    def is_empty(self):
        return len(self.vertices) == 0 or len(self.faces) == 0

    def copy(self):
        m = Mesh(self.vertices.copy(), self.faces.copy(),
                 None if self.vertex_colors is None else self.vertex_colors.copy())
        m.metadata = dict(self.metadata)
        return m

    def update_vertices(self, mask):
        mask = np.asarray(mask, bool)
        remap = np.cumsum(mask) - 1
        keep_face = mask[self.faces].all(axis=1)
        self.faces = remap[self.faces[keep_face]]
        self.vertices = self.vertices[mask]
        if self.vertex_colors is not None:
            self.vertex_colors = self.vertex_colors[mask]

    def update_faces(self, mask):
        self.faces = self.faces[np.asarray(mask)]

    def nondegenerate_faces(self):
        f = self.faces
        return (f[:, 0] != f[:, 1]) & (f[:, 1] != f[:, 2]) & (f[:, 0] != f[:, 2])

    def remove_unreferenced_vertices(self):
        used = np.zeros(len(self.vertices), bool)
        used[self.faces.reshape(-1)] = True
        self.update_vertices(used)

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
            views.append({"buffer": 0, "byteOffset": off, "byteLength": data.nbytes, "target": target})
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
        bin_blob = b"".join(chunks)
        doc = {"asset": {"version": "2.0", "generator": "r3g"}, "scene": 0, "scenes": [{"nodes": [0]}],
               "nodes": [{"mesh": 0}], "meshes": [{"primitives": [{"attributes": attrs, "indices": 1, "mode": 4}]}],
               "buffers": [{"byteLength": len(bin_blob)}], "bufferViews": views, "accessors": accessors}
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
