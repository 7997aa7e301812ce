"""Object-parallel multi-GPU plumbing of the stage: one persistent process per GPU, crops scattered
from rank 0 over RCCL (torch.distributed backend "nccl" IS RCCL on ROCm; "gloo" on the CPU-only test machines),
object indices handed out dynamically, meshes gathered back to rank 0.

The reference has no collective on this path: it spawns one process per IMAGE and exchanges files
(src/2d_to_3d_models/run.py:176-193).  Here
  * broadcast_crops : rank 0 holds the decoded RGBA crops; ONE broadcast of the packed batch (<= 64 crops x 1 MiB) puts
                      them into every rank's memory (HBM under RCCL), so any rank can take any object;
  * WorkQueue       : a shared counter in the rendezvous store (atomic add on rank 0's TCPStore): a rank claims the next
                      unprocessed object index when it becomes free -- no static i % num_devices assignment, no straggler
                      holding back objects another GPU could have taken;
  * gather_meshes   : per object (index, nV, nF) metadata by all_gather, then point-to-point send / recv of the vertex and
                      face arrays to rank 0 (variable length, a few hundred KB per cleaned mesh).
There is no all-reduce anywhere, so nothing here is ring- or bandwidth-bound: scaling is load balance only.
"""
import numpy as np
import torch
import torch.distributed as dist


def _comm_device():
    """tensors handed to collectives live in HBM under RCCL and on the host under gloo"""
    if dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def broadcast_crops(crops, src=0):
    """crops: on rank `src` a list of uint8 arrays [h, w, 4] (RGBA); ignored elsewhere.
    Returns, on every rank, the list of uint8 tensors [h, w, 4] on the communication device."""
    dev = _comm_device()
    rank = dist.get_rank()
    if rank == src:
        arrs = [np.ascontiguousarray(c, dtype=np.uint8) for c in crops]
        for a in arrs:
            if a.ndim != 3 or a.shape[2] != 4:
                raise ValueError("crops must be RGBA uint8 arrays [h, w, 4]")
        header = torch.tensor([len(arrs)] + [d for a in arrs for d in a.shape[:2]], dtype=torch.int64)
    else:
        header = None
    n = torch.zeros(1, dtype=torch.int64, device=dev)
    if rank == src:
        n[0] = header.numel()
    dist.broadcast(n, src)
    h = header.to(dev) if rank == src else torch.zeros(int(n.item()), dtype=torch.int64, device=dev)
    dist.broadcast(h, src)
    h = h.cpu().tolist()
    shapes = [(h[1 + 2 * i], h[2 + 2 * i]) for i in range(h[0])]
    total = sum(a * b * 4 for a, b in shapes)
    if rank == src:
        payload = torch.from_numpy(np.concatenate([a.reshape(-1) for a in arrs]) if arrs else np.zeros(0, np.uint8)).to(dev)
    else:
        payload = torch.empty(total, dtype=torch.uint8, device=dev)
    if total:
        dist.broadcast(payload, src)
    out, off = [], 0
    for (hh, ww) in shapes:
        out.append(payload[off:off + hh * ww * 4].view(hh, ww, 4))
        off += hh * ww * 4
    return out


class WorkQueue:
    """Dynamic hand-out of object indices 0..n-1 through the process group's store (one atomic add per claim)."""

    def __init__(self, n_items, name="r3g_queue"):
        self.n = int(n_items)
        self.key = name
        self.store = None
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            from torch.distributed import distributed_c10d as c10d
            self.store = c10d._get_default_store()
            dist.barrier()          # every rank has created its handle before the first claim
        self._local = 0

    def claim(self):
        """next unclaimed index, or None when the list is exhausted"""
        if self.store is None:
            i = self._local
            self._local += 1
        else:
            i = int(self.store.add(self.key, 1)) - 1
        return i if i < self.n else None


def gather_meshes(local, dst=0):
    """local: list of (index, vertices float32 [nV,3], faces int [nF,3]) or (index, vertices, faces, uv float32 [nV,2],
    texture uint8 [T,T,C]) tensors / arrays produced on this rank.
    Returns on rank `dst` a dict index -> (vertices float32 ndarray, faces int32 ndarray[, uv, texture]) of ALL ranks; {}
    elsewhere."""
    dev = _comm_device()
    rank, world = dist.get_rank(), dist.get_world_size()
    mine = []
    for item in local:
        idx, v, f = item[:3]
        v = torch.as_tensor(v).to(device=dev, dtype=torch.float32).contiguous().view(-1, 3)
        f = torch.as_tensor(f).to(device=dev, dtype=torch.int32).contiguous().view(-1, 3)
        uv = tex = None
        if len(item) > 3 and item[3] is not None and item[4] is not None:
            uv = torch.as_tensor(item[3]).to(device=dev, dtype=torch.float32).contiguous().view(-1, 2)
            tex = torch.as_tensor(item[4]).to(device=dev, dtype=torch.uint8).contiguous()
        mine.append((int(idx), v, f, uv, tex))
    meta = [None] * world
    dist.all_gather_object(meta, [(i, int(v.shape[0]), int(f.shape[0]), None if tex is None else tuple(tex.shape))
                                  for i, v, f, uv, tex in mine])

    def pack(v, f, uv, tex):
        return (v.cpu().numpy(), f.cpu().numpy()) if tex is None else (v.cpu().numpy(), f.cpu().numpy(), uv.cpu().numpy(),
                                                                      tex.cpu().numpy())
    out = {}
    if rank == dst:
        for i, v, f, uv, tex in mine:
            out[i] = pack(v, f, uv, tex)
        for r in range(world):
            if r == dst:
                continue
            for (i, nv, nf, tshape) in meta[r]:       # the sender walks the same list in the same order
                v = torch.empty((nv, 3), dtype=torch.float32, device=dev)
                f = torch.empty((nf, 3), dtype=torch.int32, device=dev)
                if nv:
                    dist.recv(v, src=r)
                if nf:
                    dist.recv(f, src=r)
                uv = tex = None
                if tshape is not None:
                    uv = torch.empty((nv, 2), dtype=torch.float32, device=dev)
                    tex = torch.empty(tshape, dtype=torch.uint8, device=dev)
                    if nv:
                        dist.recv(uv, src=r)
                    dist.recv(tex, src=r)
                out[i] = pack(v, f, uv, tex)
    else:
        for i, v, f, uv, tex in mine:
            if v.shape[0]:
                dist.send(v, dst=dst)
            if f.shape[0]:
                dist.send(f, dst=dst)
            if tex is not None:
                if v.shape[0]:
                    dist.send(uv, dst=dst)
                dist.send(tex, dst=dst)
    return out
