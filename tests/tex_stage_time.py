#!/usr/bin/env python3
"""Time of the texture stage at upstream's sizes with both diffusion models on the HIP blocks (random weights of the real
architectures: SD-2.1 UNet with 8 input channels + SD VAE for delighting, the 2.5D UNet + its 4-channel reference copy + SD VAE for
the six views; 50 / 30 steps at 512 x 512; 2048^2 texture, 1024^2 bake renders, a 20 480-face sphere).  Prints one JSON line.
Lives under tests/ because it takes the weights' SHAPES from the oracle's module definitions (random tensors of the real
architectures); nothing of the oracle is executed or timed.
    python tests/tex_stage_time.py > gpurun_out/r03_texture_stage_time.json"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "3d-re-gen_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    from PIL import Image
    import tex_support as ts
    from hy3dgen.texgen import Hunyuan3DPaintPipeline
    from hy3dgen.texgen.utils.dehighlight_utils import Light_Shadow_Remover
    from hy3dgen.texgen.utils.multiview_utils import Multiview_Diffusion_Net
    from oracle import aekl_torch as A, unet2p5d_torch as M, unet_torch as U      # weight SYNTHESIS only: nothing of it is timed
    from r3g.delight import InstructPix2Pix
    from r3g.mesh import Mesh
    from r3g.multiview import MultiviewPipeline, MultiviewUNet
    from r3g.unet import AutoencoderKLBlocks
    t0 = time.time()
    ucfg = dict(U.sd21_config(), in_channels=8, out_channels=4)
    vsd = A.build(A.sd_config(), seed=2).state_dict()
    delight = Light_Shadow_Remover(model=InstructPix2Pix(U.synthetic_state_dict(ucfg, seed=1, full=True), vsd, ucfg, A.sd_config(),
                                                         image_size=512),
                                   prompt_embeds=torch.randn(1, 77, 1024, generator=torch.Generator().manual_seed(0)))
    mv = MultiviewUNet(M.build(U.sd21_config(), seed=3).state_dict(), U.sd21_config(), n_views_max=6, n_ref_max=1, latent_hw=64 * 64)
    net = Multiview_Diffusion_Net(pipeline=MultiviewPipeline(mv, AutoencoderKLBlocks(vsd, max_image_hw=512 * 512)))
    setup = time.time() - t0
    v, f = ts.icosphere(5)                       # 20 480 faces (the stage hands the texture step at most 40 000)
    img = np.zeros((512, 512, 4), np.uint8)
    yy, xx = np.mgrid[0:512, 0:512]
    img[..., 0], img[..., 1], img[..., 2] = xx // 2, yy // 2, 128
    img[..., 3] = np.where((xx - 255.5) ** 2 + (yy - 255.5) ** 2 < 200 ** 2, 255, 0)
    image = Image.fromarray(img, "RGBA")
    pipe = Hunyuan3DPaintPipeline(texture_size=2048, render_size=1024, multiview_model=net, delight_model=delight)
    times = {}

    def timed(name, fn):
        torch.cuda.synchronize()
        t = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        times[name] = round(1000.0 * (time.perf_counter() - t), 1)
        return out

    mesh = Mesh(v, f)
    timed("warm_up_whole_stage_ms", lambda: pipe(mesh, image=image))
    out = timed("whole_stage_ms", lambda: pipe(mesh, image=image))
    timed("delight_only_ms", lambda: delight(image))
    nm = [Image.new("RGB", (512, 512), (128, 128, 255))] * 6
    timed("multiview_only_ms", lambda: net(image.convert("RGB"), nm + nm, [21, 12, 15, 18, 43, 37]))
    # round 6: the same two flows once more with every launch bracketed by HIP events (r3g_prof_*): algorithmic
    # FLOPs and achieved TFLOP/s of the MFMA families (every convolution / linear layer is a GEMM launch, attention is the flash
    # kernel), milliseconds of the rest
    import ctypes
    from r3g import ffi
    L = ffi.lib()
    fams = ["gemm", "attention", "layernorm", "qkv_split", "gemv", "elementwise", "mc_classify", "mc_other", "mesh"]
    roof = {}
    for name, fn in (("delight", lambda: delight(image)), ("multiview", lambda: net(image.convert("RGB"), nm + nm, [21, 12, 15, 18, 43, 37]))):
        torch.cuda.synchronize()
        ffi.check(L.r3g_prof_enable(1))
        t = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        wall = 1000.0 * (time.perf_counter() - t)
        n = len(fams)
        cnt, ms, work = (ctypes.c_int64 * n)(), (ctypes.c_double * n)(), (ctypes.c_double * n)()
        ffi.check(L.r3g_prof_read(cnt, ms, work, n))
        ffi.check(L.r3g_prof_enable(0))
        roof[name] = {"wall_ms_with_events": round(wall, 1),
                      "families": {fams[i]: {"launches": int(cnt[i]), "ms": round(float(ms[i]), 2),
                                             **({"tflop": round(float(work[i]) / 1e12, 3),
                                                 "tflops_achieved": round(float(work[i]) / max(float(ms[i]), 1e-9) / 1e9, 1),
                                                 "frac_of_2500": round(float(work[i]) / max(float(ms[i]), 1e-9) / 1e9 / 2500.0, 3)} if fams[i] in ("gemm", "attention") else {})}
                                   for i in range(n) if cnt[i]}}
    times["roofline"] = roof
    print(json.dumps({"what": "texture stage at upstream's sizes, both diffusion models on the HIP blocks, random weights",
                      "faces": int(len(f)), "texture": list(out.texture.shape), "delight_steps": delight.steps, "multiview_steps": net.steps,
                      "views": 6, "view_size": net.view_size, "guidance_scale": 2.0, "setup_seconds_cpu_weight_synthesis": round(setup, 1),
                      "times_ms": times, "source": out.metadata["texture_source"], "stats": pipe.last_stats}))


if __name__ == "__main__":
    main()
