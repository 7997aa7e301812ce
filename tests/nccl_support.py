"""TEST INFRASTRUCTURE for tests/workers/nccl_one_rank.py: the stage's factory on a tiny seeded model (oracle's tiny config;
real HIP pipeline, real cleaners, texture stage in its input-view mode), so that run_distributed finishes in seconds."""
import torch


def tiny_factory(config, device):
    from hy3dgen.shapegen import DegenerateFaceRemover, FaceReducer, FloaterRemover, Hunyuan3DDiTFlowMatchingPipeline
    from hy3dgen.texgen import Hunyuan3DPaintPipeline
    from oracle import hy3d_torch as H
    cfg = H.tiny_config()
    sd = {k: (t.to(torch.bfloat16).float() if t.ndim >= 2 else t) for k, t in H.synthetic_state_dict(cfg, seed=5).items()}
    shapegen = Hunyuan3DDiTFlowMatchingPipeline(cfg, sd, device, grid_chunk=2048)
    texgen = Hunyuan3DPaintPipeline(texture_size=256, render_size=128)
    return shapegen, texgen, [FloaterRemover(), DegenerateFaceRemover(), FaceReducer()]
