"""TEST INFRASTRUCTURE: CPU stand-ins for the four places where the hy3dgen mirror touches the GPU, so that the
reference's own stage script can be executed, unmodified, in the GPU-less build container (tests/test_reference_script.py).

What stays the product's code on this run: the whole import surface of hy3dgen.{shapegen,texgen,rembg}, from_pretrained
(snapshot directory, config.yaml, safetensors), ImageProcessorV2 / conditioner transform, the pipeline's __call__ (argument
handling, CFG / scheduler plumbing, grid -> mesh), the Trimesh-like Mesh class, the cleaner classes' call surface, the GLB
writer.  What is replaced: the device model (by the fp32 oracle), marching cubes (by the C oracle) and the mesh cleaners'
kernels (by the numpy restatement / the host run of the edge-collapse code), and the texture-stage primitives of r3g.texops
(by oracle/tex_ref.py; the texgen pipeline object, its view set-up, unwrap and image registration stay the product's)."""
import contextlib

import numpy as np
import torch


class OracleModel:
    """the interface of r3g.model.ShapeModel on top of oracle.hy3d_torch (CPU, fp32)"""

    def __init__(self, cfg, state_dict, grid_chunk=0):
        from oracle import hy3d_torch as H
        self.cfg = cfg
        sd = {k: (v.float() if torch.is_floating_point(v) else v) for k, v in state_dict.items()}
        self.pipe = H.load_state_dict(H.ShapePipeline(cfg), sd)
        self.H = H
        p = cfg["cond"]["image_size"] // cfg["cond"]["patch_size"]
        self.cond_tokens = p * p + 1
        self.num_latents = cfg["vae"]["num_latents"]
        self.in_channels = cfg["dit"]["in_channels"]
        self._z = None

    @torch.no_grad()
    def cond_encode(self, image):
        return self.pipe.conditioner.main_image_encoder.model(image[None].float()).last_hidden_state[0].to(torch.bfloat16)

    @torch.no_grad()
    def flow_sample(self, latents, cond2, steps, guidance_scale, shift=1.0, uncond_uniform=None):
        return self.pipe.sample(cond2.float(), latents[None].float(), steps, guidance_scale)[0]

    @torch.no_grad()
    def vae_decode(self, latents, return_z=False):
        self._z = self.pipe.vae(latents[None].float() / self.pipe.vae.scale_factor)
        return self._z[0] if return_z else None

    @torch.no_grad()
    def grid_query(self, bound, octree_resolution, out=None, start=0, count=None):
        return self.H.volume_decode(self.pipe.vae, self._z, bound, octree_resolution, 4096)


def install():
    import hy3dgen.shapegen as sg
    import hy3dgen.shapegen.pipelines as pl
    from oracle import mc as omc
    from oracle import mesh_clean
    from r3g import mesh as rmesh
    from r3g import meshops
    import emu_qem

    class CpuPipeline(pl.Hunyuan3DDiTFlowMatchingPipeline):
        def __init__(self, cfg, state_dict, device="cpu", grid_chunk=0):
            super().__init__(cfg, state_dict, "cuda:0", grid_chunk)
            self.device = torch.device("cpu")

        def _make_model(self, cfg, state_dict, grid_chunk):
            return OracleModel(cfg, state_dict, grid_chunk)

        def _device_ctx(self):
            return contextlib.nullcontext()

        def _extract_mesh(self, grid, mc_level, box_v, octree_resolution):
            v, f = omc.hy3d_mesh(grid.numpy(), mc_level, box_v, octree_resolution)
            return torch.from_numpy(v), torch.from_numpy(f.astype(np.int32))

    pl.Hunyuan3DDiTFlowMatchingPipeline = CpuPipeline
    sg.Hunyuan3DDiTFlowMatchingPipeline = CpuPipeline

    def host(fn):
        def run(verts, faces, *args):
            v, f = fn(verts.numpy(), faces.numpy(), *args)[:2]
            return torch.from_numpy(np.ascontiguousarray(v, np.float32)), torch.from_numpy(np.ascontiguousarray(f, np.int32))
        return run
    meshops.remove_floaters = host(mesh_clean.remove_floaters)
    meshops.remove_degenerate = host(mesh_clean.remove_degenerate)
    meshops.reduce_faces = host(emu_qem.reduce_faces)

    def device_buffers(self, device=None):
        if self._dv is None:
            self._dv = torch.from_numpy(np.ascontiguousarray(self._v, np.float32))
            self._df = torch.from_numpy(np.ascontiguousarray(self._f, np.int32))
        return self._dv, self._df
    rmesh.Mesh.device_buffers = device_buffers

    # texture-stage primitives: numpy restatement behind the r3g.texops call surface, tensors on the host
    from oracle import tex_ref
    from r3g import texops
    import hy3dgen.texgen.pipelines as tp

    def n(t):
        return t.detach().cpu().numpy()

    def rasterize(pos_clip, tri, height, width):
        fi, bary = tex_ref.rasterize(n(pos_clip), n(tri), height, width)
        return torch.from_numpy(fi), torch.from_numpy(bary)
    texops.rasterize = rasterize
    texops.interpolate = lambda attr, tri, fi, bary: torch.from_numpy(tex_ref.interpolate(n(attr), n(tri), n(fi), n(bary)))
    texops.view_weight = lambda fi, depth, normal, cos_threshold=0.1, depth_edge=0.01, view_weight=1.0, power=4.0: \
        torch.from_numpy(tex_ref.view_weight(n(fi), n(depth), n(normal), cos_threshold, depth_edge, view_weight, power))
    texops.new_accumulator = lambda tex_size, device: torch.zeros((tex_size, tex_size, 4), dtype=torch.int64)

    def bake(image, weight, fi, bary, uv, uv_tri, acc):
        a = n(acc).view(np.uint64)
        tex_ref.bake(n(image), n(weight), n(fi), n(bary), n(uv), n(uv_tri), a.shape[0], a)
        acc.copy_(torch.from_numpy(a.view(np.int64)))
        return acc
    texops.bake = bake

    def bake_gather(fi_uv, bary_uv, clip_uv, uv_tri, image, weight, fi, depth, acc, depth_eps=0.01):
        a = n(acc).view(np.uint64)
        tex_ref.bake_gather(n(fi_uv), n(bary_uv), n(clip_uv), n(uv_tri), n(image), n(weight), n(fi), n(depth), depth_eps, a)
        acc.copy_(torch.from_numpy(a.view(np.int64)))
        return acc
    texops.bake_gather = bake_gather

    def bake_finalize(acc):
        tex, mask = tex_ref.bake_finalize(n(acc).view(np.uint64))
        return torch.from_numpy(tex), torch.from_numpy(mask)
    texops.bake_finalize = bake_finalize

    def inpaint(texture, mask, fi_uv, bary_uv, verts, pos_tri, uv, uv_tri, dilate_iters=8):
        tex, m, rounds = tex_ref.inpaint(n(texture), n(mask), n(fi_uv), n(bary_uv), n(verts), n(pos_tri), n(uv), n(uv_tri), dilate_iters)
        return torch.from_numpy(tex), torch.from_numpy(m), rounds
    texops.inpaint = inpaint
    tp.Hunyuan3DPaintPipeline._device = lambda self: torch.device("cpu")
