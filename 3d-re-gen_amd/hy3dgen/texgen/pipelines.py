"""hy3dgen.texgen.pipelines -- MI355X mirror of upstream's texture pipeline entry point.

Reference call sites: src/2d_to_3d_models/run.py:17 (import), :126-128 / :207-209 (`from_pretrained`), :97
(`mesh = pipeline_texgen(mesh, image=image)`).  Upstream's __call__ (recalled): delight the input image with an
image-to-image diffusion model -> UV-unwrap the mesh (xatlas) -> render normal / position maps of six fixed views with the
native `custom_rasterizer` -> generate the six views with a multiview diffusion UNet -> back-project and blend them into a
UV texture -> inpaint what no view saw (`mesh_processor.meshVerticeInpaint` + cv2) -> textured mesh.

What exists here: every step, the two diffusion models included -- but no checkpoint for either, and what
is recalled of upstream's code is not pinned against it.  The delighting model (`delight_model=`:
hy3dgen.texgen.utils.dehighlight_utils.Light_Shadow_Remover -- SD-2.x UNet, SD VAE and Euler-ancestral loop on the HIP blocks) runs
when the caller has its checkpoint; without one the input image is used as it is.  The native pieces run as HIP
kernels behind the C ABI (r3g.texops: rasterise, interpolate, view weights, fixed-point baking, vertex-propagation
inpainting); the unwrap is r3g.uvatlas.chart_atlas (axis-projected height-field charts at uniform texel density; the per-face
atlas of rounds 1-2 remains as `atlas="face"`).  The views that get baked are
  * the views a `hy3dgen.texgen.utils.multiview_utils.Multiview_Diffusion_Net` generates from the (delighted) image and the
    normal / position maps this pipeline renders for the six views -- upstream's flow, on the HIP 2.5D UNet -- when the caller has
    that model's checkpoint, or
  * the views a caller-supplied `multiview_model(image, views) -> list of RGB images` produces, when one is given, or
  * by default ONLY the input image, registered to the front view of the mesh (its alpha silhouette against the mesh
    silhouette); every texel no view saw is filled by colour propagation over the mesh.
`self.source` says which, and the stage report repeats it: a GLB textured from one view is NOT what upstream produces."""
import os

import numpy as np

# (elevation, azimuth, blend weight) of upstream's six candidate views.  [UPSTREAM-RECALLED] Hunyuan3DTexGenConfig:
# candidate_camera_azims = [0, 90, 180, 270, 0, 180], candidate_camera_elevs = [0, 0, 0, 0, 90, -90],
# candidate_view_weights = [1, 0.1, 0.5, 0.1, 0.05, 0.05] -- the view from below is taken at azimuth 180 (camera index 37 of
# the multiview model; rounds 2-3 had it at azimuth 0 = index 39).  Unpinned: to be checked against the public source.
DEFAULT_VIEWS = [(0, 0, 1.0), (0, 90, 0.1), (0, 180, 0.5), (0, 270, 0.1), (90, 0, 0.05), (-90, 180, 0.05)]


def view_rotation(elev_deg, azim_deg):
    """world -> camera rotation: azimuth about +y, then elevation about +x; the camera looks down -z, +y is up"""
    a, e = np.deg2rad(azim_deg), np.deg2rad(elev_deg)
    ry = np.array([[np.cos(a), 0, -np.sin(a)], [0, 1, 0], [np.sin(a), 0, np.cos(a)]])
    rx = np.array([[1, 0, 0], [0, np.cos(e), -np.sin(e)], [0, np.sin(e), np.cos(e)]])
    return (rx @ ry).astype(np.float32)


def ortho_clip(cam_xyz, centre, half_extent, depth_half):
    """orthographic clip coordinates: x right, image row 0 at the top (+y up), nearer = smaller depth"""
    out = np.ones((len(cam_xyz), 4), np.float32)
    out[:, 0] = (cam_xyz[:, 0] - centre[0]) / half_extent[0]
    out[:, 1] = -(cam_xyz[:, 1] - centre[1]) / half_extent[1]
    out[:, 2] = -cam_xyz[:, 2] / depth_half
    return out


class Hunyuan3DPaintPipeline:
    implemented = True

    def __init__(self, texture_size=None, render_size=None, multiview_model=None, views=None, cos_threshold=0.1,
                 depth_edge=0.02, power=4.0, dilate_iters=8, device=None, atlas=None, delight_model=None):
        # [UPSTREAM-RECALLED] Hunyuan3DTexGenConfig: texture_size 2048, render_size 2048; R3G_TEX_SIZE / R3G_TEX_RENDER override
        # the defaults (tests).  (Rounds 2-3 rendered at 1024.)
        self.texture_size = int(texture_size or os.environ.get("R3G_TEX_SIZE", 2048))
        self.render_size = int(render_size or os.environ.get("R3G_TEX_RENDER", 2048))
        self.multiview_model = multiview_model
        # upstream: self.models['delight_model'] = Light_Shadow_Remover(config), applied to the image first thing in __call__;
        # here an instance of hy3dgen.texgen.utils.dehighlight_utils.Light_Shadow_Remover (HIP UNet + VAE) when the caller has one
        self.delight_model = delight_model
        self.views = list(views) if views is not None else list(DEFAULT_VIEWS)
        self.cos_threshold, self.depth_edge, self.power = float(cos_threshold), float(depth_edge), float(power)
        self.dilate_iters = int(dilate_iters)
        self.atlas = atlas or os.environ.get("R3G_TEX_ATLAS", "chart")     # "chart" (default) | "face"
        if self.atlas not in ("chart", "face"):
            raise ValueError("atlas must be 'chart' or 'face'")
        self.device = device
        self.last_stats = {}

    @property
    def source(self):
        pre = "delighted input; " if self.delight_model is not None else ""
        if self.multiview_model is not None and getattr(self.multiview_model, "wants_control_images", False):
            return pre + "multiview diffusion model (normal / position maps of %d views), all views baked" % len(self.views)
        if self.multiview_model is not None:
            return pre + "multiview model supplied by the caller, %d views baked" % len(self.views)
        return pre + "input view only (no multiview diffusion model on this path); unseen texels filled by propagation over the mesh"

    DELIGHT_SUBFOLDER = "hunyuan3d-delight-v2-0"      # upstream's from_pretrained: <model_path>/<these two folders>
    MULTIVIEW_SUBFOLDER = "hunyuan3d-paint-v2-0"

    @classmethod
    def from_pretrained(cls, model_path=None, subfolder=None, **kwargs):
        """upstream builds Hunyuan3DTexGenConfig(<model_path>/hunyuan3d-delight-v2-0, <model_path>/hunyuan3d-paint-v2-0) and loads
        the two diffusion models from there.  Here: a folder that exists is loaded (a folder that cannot be loaded is an error;
        with strict=False it is recorded in `load_problems` and the pipeline runs without that model -- the stage does that, loudly,
        so that a snapshot whose texture folders this loader cannot read still yields meshes); a model passed explicitly wins;
        with neither, the pipeline runs without that model and `source` says so.  The models live on `device` (default: the
        process's current device -- a rank's own GPU)"""
        allowed = ("texture_size", "render_size", "multiview_model", "views", "cos_threshold", "depth_edge", "power",
                   "dilate_iters", "device", "atlas", "delight_model")
        kw = {k: v for k, v in kwargs.items() if k in allowed}
        strict = bool(kwargs.get("strict", True))
        if model_path and os.path.isdir(str(model_path)):
            dev = kw.get("device")
            if dev is None:                       # the process's current device (a rank's own GPU), as _device() chooses it
                import torch
                dev = torch.cuda.current_device() if torch.cuda.is_available() else 0
            if not isinstance(dev, int):          # "cuda:1" / torch.device -> the index the r3g classes take
                if ":" in str(dev):
                    dev = int(str(dev).rsplit(":", 1)[1])
                else:
                    import torch
                    dev = torch.cuda.current_device() if torch.cuda.is_available() else 0
            kw["device"] = "cuda:%d" % dev

            class _Cfg:
                device = dev
            d = os.path.join(str(model_path), cls.DELIGHT_SUBFOLDER)
            m = os.path.join(str(model_path), cls.MULTIVIEW_SUBFOLDER)
            problems = []

            def load(what, folder, make):
                if kw.get(what) is not None or not os.path.isdir(folder):
                    return
                try:
                    kw[what] = make()
                except Exception as e:      # noqa: BLE001 -- a checkpoint folder this loader cannot read
                    if strict:
                        raise
                    problems.append("%s: %s: %s" % (os.path.basename(folder), type(e).__name__, e))

            def make_delight():
                from .utils.dehighlight_utils import Light_Shadow_Remover
                _Cfg.light_remover_ckpt_path = d
                return Light_Shadow_Remover(_Cfg)

            def make_multiview():
                from .utils.multiview_utils import Multiview_Diffusion_Net
                _Cfg.multiview_ckpt_path = m
                return Multiview_Diffusion_Net(_Cfg)
            load("delight_model", d, make_delight)
            load("multiview_model", m, make_multiview)
            pipe = cls(**kw)
            pipe.load_problems = problems      # strict=False: what could not be loaded (the stage prints it and goes on)
            return pipe
        return cls(**kw)

    # -- helpers -----------------------------------------------------------------------------------------------------
    def _device(self):
        import torch
        return torch.device(self.device or ("cuda:%d" % torch.cuda.current_device()))

    @staticmethod
    def recenter_image(image, border_ratio=0.2):
        """[UPSTREAM-RECALLED] Hunyuan3DPaintPipeline.recenter_image, the first thing upstream's __call__ does with the image prompt:
        crop an RGBA image to its alpha bounding box and paste it, with a border of `border_ratio` of the crop on every side, in the
        middle of a transparent square; RGB / L images pass through"""
        from PIL import Image
        if image.mode == "RGB":
            return image
        if image.mode == "L":
            return image.convert("RGB")
        if image.mode != "RGBA":
            image = image.convert("RGBA")
        alpha = np.asarray(image)[:, :, 3]
        rows, cols = np.nonzero(alpha > 0)
        if len(rows) == 0:
            raise ValueError("Image is fully transparent")
        cropped = image.crop((int(cols.min()), int(rows.min()), int(cols.max()) + 1, int(rows.max()) + 1))
        width, height = cropped.size
        bw, bh = int(width * border_ratio), int(height * border_ratio)
        new_w, new_h = width + 2 * bw, height + 2 * bh
        side = max(new_w, new_h)
        out = Image.new("RGBA", (side, side), (255, 255, 255, 0))
        out.paste(cropped, ((side - new_w) // 2 + bw, (side - new_h) // 2 + bh))
        return out

    def _delight(self, image):
        """upstream: image_prompt = self.models['delight_model'](image_prompt).  The delighted picture comes back as RGB over
        white at the model's size; the alpha it is registered by stays the input's."""
        from PIL import Image
        out = self.delight_model(image)
        if out.mode != "RGBA" and image.mode == "RGBA":
            out = out.convert("RGB")
            out.putalpha(image.getchannel("A").resize(out.size, Image.BILINEAR))
        return out

    def _control_maps(self, v, corner_n, views, size, radius, dev, d_f, d_ct):
        """normal maps and position maps of the views, as upstream's render_normal_multiview(use_abs_coor=True) /
        render_position_multiview feed the multiview model: world-space normals as (n + 1) / 2, positions scaled into [0, 1],
        white where the view sees no surface ([UPSTREAM-RECALLED]: the exact colour conventions are not pinned)"""
        import torch
        from PIL import Image
        from r3g import texops
        d_v = torch.from_numpy(np.ascontiguousarray(v, np.float32)).to(dev)
        d_n = torch.from_numpy(np.ascontiguousarray(corner_n, np.float32)).to(dev)
        nmaps, pmaps = [], []
        for (elev, azim, _) in views:
            cam = v @ view_rotation(elev, azim).T
            clip = torch.from_numpy(ortho_clip(cam, (0.0, 0.0), np.array([radius, radius], np.float32), radius * 2.0)).to(dev)
            fi, bary = texops.rasterize(clip, d_f, size, size)
            seen = (fi > 0)[..., None]
            n_img = torch.where(seen, texops.interpolate(d_n, d_ct, fi, bary) * 0.5 + 0.5, torch.ones((), device=dev))
            p_img = torch.where(seen, texops.interpolate(d_v, d_f, fi, bary) / (2.0 * radius) + 0.5, torch.ones((), device=dev))
            for dst, img in ((nmaps, n_img), (pmaps, p_img)):
                dst.append(Image.fromarray((img.clamp(0, 1) * 255.0 + 0.5).to(torch.uint8).cpu().numpy(), "RGB"))
        return nmaps, pmaps

    def _front_image(self, image):
        """RGBA PIL image -> (rgb float32 [R, R, 3] in [0, 1], alpha float32 [R, R], alpha bbox in [0, 1] image coordinates)"""
        from PIL import Image
        r = self.render_size
        img = image.convert("RGBA") if image.mode != "RGBA" else image
        arr = np.asarray(img.resize((r, r), Image.BILINEAR), np.float32) / 255.0
        alpha = arr[..., 3]
        ys, xs = np.nonzero(alpha > 0.5)
        if len(xs) == 0:
            box = (0.0, 0.0, 1.0, 1.0)
        else:
            box = (xs.min() / r, ys.min() / r, (xs.max() + 1) / r, (ys.max() + 1) / r)
        return np.ascontiguousarray(arr[..., :3]), np.ascontiguousarray(alpha), box

    def __call__(self, mesh, image=None, **kwargs):
        import torch
        from r3g import texops, uvatlas
        from r3g.mesh import Mesh
        if image is None:
            raise ValueError("Hunyuan3DPaintPipeline needs the object's image")
        if mesh.is_empty:
            return mesh
        upstream_flow = self.delight_model is not None or getattr(self.multiview_model, "wants_control_images", False)
        if upstream_flow:
            image = self.recenter_image(image)
        if self.delight_model is not None:
            image = self._delight(image)
        dev = self._device()
        v = np.ascontiguousarray(mesh.vertices, np.float32)
        f = np.ascontiguousarray(mesh.faces, np.int32)
        nf = len(f)
        T, R = self.texture_size, self.render_size
        if self.atlas == "chart":
            try:
                uv, uv_tri, uv_to_pos, chart = uvatlas.chart_atlas(v, f, T)
                n_charts = int(chart.max()) + 1
            except ValueError:          # more charts than the texture has room for: every face its own cell
                uv = None
        else:
            uv = None
        if uv is None:
            uv, uv_tri = uvatlas.face_atlas(nf, T)
            uv_to_pos, n_charts = f.reshape(-1), nf
        corner_tri = np.arange(3 * nf, dtype=np.int32).reshape(nf, 3)             # per-corner attributes (flat normals)
        e1, e2 = v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]]
        fn = np.cross(e1, e2)
        fn /= np.maximum(np.linalg.norm(fn, axis=1, keepdims=True), 1e-20)
        corner_n = np.repeat(fn.astype(np.float32), 3, axis=0)                    # flat normals, one per face corner

        d_f, d_uv, d_uvt = torch.from_numpy(f).to(dev), torch.from_numpy(uv).to(dev), torch.from_numpy(uv_tri).to(dev)
        d_ct = torch.from_numpy(corner_tri).to(dev)
        rgb, alpha, abox = self._front_image(image)
        if self.multiview_model is not None and getattr(self.multiview_model, "wants_control_images", False):
            # upstream's call: multiview_model(image_prompt, normal_maps + position_maps, camera_info)
            from PIL import Image
            from .utils.multiview_utils import camera_index
            size = int(getattr(self.multiview_model, "view_size", 512))
            radius0 = float(np.abs(v).max()) * 1.05 + 1e-6
            nmaps, pmaps = self._control_maps(v, corner_n, self.views, size, radius0, dev, d_f, d_ct)
            cams = [camera_index(e, a) for (e, a, _) in self.views]
            out = self.multiview_model(image, nmaps + pmaps, cams)
            images = [np.ascontiguousarray(np.asarray(im.convert("RGB").resize((R, R), Image.BILINEAR), np.float32) / 255.0)
                      for im in out]
            views = self.views
        elif self.multiview_model is not None:
            images = [np.ascontiguousarray(np.asarray(im, np.float32)) for im in self.multiview_model(image, self.views)]
            views = self.views
        else:
            images, views = [rgb], [(0, 0, 1.0)]
        radius = float(np.abs(v).max()) * 1.05 + 1e-6

        # the mesh in UV space: which face and where inside it every texel is
        fi_uv, bary_uv = texops.rasterize(torch.from_numpy(uvatlas.uv_clip(uv)).to(dev), d_uvt, T, T)

        def bake_views(sign):
            acc = texops.new_accumulator(T, dev)
            for (elev, azim, vw), img in zip(views, images):
                rot = view_rotation(elev, azim)
                cam = v @ rot.T
                if self.multiview_model is None:
                    # register the mesh silhouette's bounding box with the alpha bounding box of the input image
                    lo, hi = cam[:, :2].min(axis=0), cam[:, :2].max(axis=0)
                    size = np.maximum(hi - lo, 1e-6)
                    wx, wy = max(abox[2] - abox[0], 1e-3), max(abox[3] - abox[1], 1e-3)
                    half = np.array([size[0] / wx, size[1] / wy], np.float32) * 0.5
                    cx = lo[0] + size[0] * 0.5 - ((abox[0] + abox[2]) * 0.5 - 0.5) * 2.0 * half[0]
                    cy = lo[1] + size[1] * 0.5 + ((abox[1] + abox[3]) * 0.5 - 0.5) * 2.0 * half[1]
                    centre = (cx, cy)
                else:
                    half, centre = np.array([radius, radius], np.float32), (0.0, 0.0)
                clip = ortho_clip(cam, centre, half, radius * 2.0)
                d_clip = torch.from_numpy(clip).to(dev)
                fi, bary = texops.rasterize(d_clip, d_f, R, R)
                depth = texops.interpolate(d_clip[:, 2:3].contiguous(), d_f, fi, bary)[..., 0]
                nmap = texops.interpolate(torch.from_numpy(np.ascontiguousarray(sign * (corner_n @ rot.T), np.float32)).to(dev),
                                          d_ct, fi, bary)
                w = texops.view_weight(fi, depth, nmap, self.cos_threshold, self.depth_edge, vw, self.power)
                if self.multiview_model is None:
                    w = w * torch.from_numpy((alpha > 0.5).astype(np.float32)).to(dev)   # only what the image actually shows
                if img.shape[0] != R or img.shape[1] != R:
                    raise ValueError("view images must be %d x %d" % (R, R))
                # texel-centric: every covered texel looks itself up in this view (clip position of every UV vertex)
                texops.bake_gather(fi_uv, bary_uv, torch.from_numpy(np.ascontiguousarray(clip[uv_to_pos])).to(dev), d_uvt,
                                   torch.from_numpy(img).to(dev), w, fi, depth, acc, self.depth_edge)
            return texops.bake_finalize(acc)

        tex, mask = bake_views(1.0)
        if int((mask > 0).sum()) == 0:
            tex, mask = bake_views(-1.0)       # a mesh wound the other way round: its normals point inwards
        painted = int((mask > 0).sum())
        tex, mask, rounds = texops.inpaint(tex, mask, fi_uv, bary_uv, torch.from_numpy(v).to(dev), d_f, d_uv, d_uvt,
                                           self.dilate_iters)
        covered = int((fi_uv > 0).sum())
        self.last_stats = {"texels_covered": covered, "texels_painted_by_views": painted,
                           "texels_coloured": int((mask > 0).sum()), "propagation_rounds": rounds, "source": self.source,
                           "atlas": self.atlas, "charts": n_charts, "uv_vertices": int(len(uv))}
        tex8 = (tex.clamp(0, 1) * 255.0 + 0.5).to(torch.uint8).cpu().numpy()
        out = Mesh(v[uv_to_pos], uv_tri, uv=uv, texture=tex8)     # a vertex is duplicated only where charts meet
        out.metadata = dict(getattr(mesh, "metadata", {}))
        out.metadata["texture_source"] = self.source
        return out
