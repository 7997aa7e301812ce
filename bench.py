#!/usr/bin/env python3
"""bench.py -- objects/sec of the Hunyuan_2d_to_3d hot path on MI355X (BASELINE.json's metric).

    python bench.py --gpus 1 --steps 8 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A "step" is ONE object: a synthetic 512x512 RGBA crop (host PIL image, as the segmentation stage hands it over)
-> preprocess -> DINOv2-g conditioner -> 50-step flow-matching DiT with CFG (batch 2) -> shape-VAE decode ->
dense 257^3 occupancy-grid query -> Lewiner marching cubes, mesh left in HBM.  The crops of a scene are independent:
--objects-per-launch of them (default 4) share the launches of the denoising loop (upstream's batch dimension; each
object's result is bit-identical to its single-object run, tests/test_model_gpu.py), everything else runs per object.  Model load, mesh cleaners,
texture generation and GLB export are outside the metric.  Workload at N=1 = BASELINE.json
configs[1] ("1 scene / 8 object crops, Hunyuan3D-2 base bf16, 50 steps, 256^3 grid"): the default --steps 8 is
one scene.  Weights are seeded synthetic (no checkpoint / network here); the arithmetic is the full model's.

At N > 1 (one rank per GPU over RCCL) rank 0 generates the N*K crops and broadcasts the packed batch into every rank's
HBM, rank r processes crops r, r+N, ... (K objects per GPU: "scaling": "weak") and the raw meshes are gathered back to
rank 0 inside the timed region (r3g/dist.py; the reference uses a process pool and the filesystem,
src/2d_to_3d_models/run.py:176-193): value = N*K / max-over-ranks time.  A second, shorter measurement with a FIXED
total of 8 crops (configs[1]'s scene split over the N GPUs, objects claimed dynamically from the shared queue) is
reported as "strong".

Rank 0 prints one JSON line.  Besides the contract fields it carries
  roofline     : the dominant kernel family (bf16 MFMA GEMM), timed live with HIP events on the launch stream
                 over one further object: achieved = sum of algorithmic FLOPs / sum of launch durations
  roofline_mc  : the HBM-bound kernel of the path (marching cubes): algorithmic bytes 4 (R+1)^3 + 12 V + 12 F over its
                 measured time, against the 8 TB/s HBM peak
  cpu_baseline : the PyTorch-CPU oracle (fp32 = `value`, and bf16 under autocast beside it) + the C marching-cubes oracle timed on
                 this box's host cores on a bounded sample of the same workload and extrapolated (rank 0, N=1 only)
  mc_parity    : the last timed object's mesh (faces AND float32 vertices) against the C oracle's mesh of the same grid,
                 computed for the CPU baseline anyway: "exact" or the run fails
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "3d-re-gen_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBPS = 8000.0
# HBM-side traffic of the dominant family per launch comes from a rocprofv3 PMC collection (FETCH_SIZE and WRITE_SIZE in
# separate passes, FETCH_SIZE doubled: gfx950 counts 128-B requests at 64 B for 16-byte loads, MI355X_MICROARCH.md "HBM")
# recorded in profiles/traffic.json together with the commit it was measured on; it is not measured by this run.
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "traffic.json")


def recorded_traffic(family):
    """the PMC record of a kernel family + whether it was measured on the library this run loaded (source digest)"""
    try:
        with open(TRAFFIC_FILE) as f:
            rec = json.load(f)
        fam = rec.get(family)
        if fam is None:
            return None
        fam = dict(fam)
        try:
            with open(os.path.join(ROOT, "3d-re-gen_amd", "libr3g.digest")) as f:
                now = f.read().strip()
        except OSError:
            now = None
        fam["stale"] = not (now and rec.get("library_digest") == now)
        return fam
    except (OSError, ValueError):
        return None


FAMILIES =["gemm", "attention", "layernorm", "qkv_split", "gemv", "elementwise", "mc_classify", "mc_other", "mesh"]


def synthetic_crop(i, size=512):
    """SURVEY.md 8d synthetic crop i: seeded alpha blob (35-65 % coverage) + smooth colour noise on white."""
    from PIL import Image, ImageDraw, ImageFilter
    rng = np.random.default_rng(1000 + i)
    alpha = Image.new("L", (size, size), 0)
    draw = ImageDraw.Draw(alpha)
    for _ in range(int(rng.integers(3, 7))):
        cx, cy = rng.uniform(0.3, 0.7, 2) * size
        w, h = rng.uniform(0.35, 0.65, 2) * size
        box = [cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2]
        if rng.random() < 0.5:
            draw.ellipse(box, fill=255)
        else:
            draw.rounded_rectangle(box, radius=float(rng.uniform(5, 40)), fill=255)
    alpha = alpha.filter(ImageFilter.GaussianBlur(2))
    low = (rng.uniform(0, 255, (8, 8, 3))).astype(np.uint8)
    rgb = np.asarray(Image.fromarray(low, "RGB").resize((size, size), Image.BILINEAR)).copy()
    a = np.asarray(alpha)
    rgb[a == 0] = 255
    return Image.fromarray(np.dstack([rgb, a]), "RGBA")


def flops_per_object(cfg, steps, R):
    """SURVEY.md 8d formulae (K/V projection of the geo decoder counted once)."""
    d, v = cfg["dit"], cfg["vae"]
    H, m = d["hidden_size"], d["mlp_ratio"]
    Lc = (cfg["cond"]["image_size"] // cfg["cond"]["patch_size"]) ** 2 + 1
    T = v["num_latents"] + Lc
    dbl = 2 * T * (3 * H * H + H * H + 2 * m * H * H) + 4 * T * T * H
    sgl = 2 * T * (H * (3 * H + m * H) + (H + m * H) * H) + 4 * T * T * H
    f_dit = d["depth"] * dbl + d["depth_single_blocks"] * sgl
    W, N = v["width"], v["num_latents"]
    f_vae = v["num_decoder_layers"] * (2 * N * (3 * W * W + W * W + 8 * W * W) + 4 * N * N * W)
    f_q = 2 * (51 * W + 2 * W * W + 2 * 4 * W * W + W) + 4 * N * W
    return 2 * steps * f_dit + f_vae + (R + 1) ** 3 * f_q


HOST_THREADS = torch.get_num_threads()   # (256 hardware threads oversubscribe the fp32 oracle:
                                         # the same sample took 96 s instead of 12 s, profiles/r04_bench_cpu256_threads.json)


def cpu_baseline(cfg, steps, R, grid_np):
    """Oracle (PyTorch CPU restatement + C marching cubes) on a bounded sample, extrapolated to one object: fp32 (the parity
    oracle's own arithmetic; `value`) and bf16 beside it (the same modules under torch.autocast("cpu", bfloat16): matrix products and
    attention in bf16, norms / softmax statistics in fp32 -- what an AMX / AVX512-bf16 host runs fastest)."""
    from oracle import hy3d_torch as H
    from oracle import mc as omc
    torch.set_num_threads(HOST_THREADS)             # torch's own choice for this host (its physical cores), whatever ran before
    torch.manual_seed(0)
    wide = H.wide_config(depth=1, depth_single=1, vae_layers=1, cond_layers=1)
    pipe = H.ShapePipeline(wide)
    d, v = cfg["dit"], cfg["vae"]
    Lc = (cfg["cond"]["image_size"] // cfg["cond"]["patch_size"]) ** 2 + 1
    x = torch.randn(2, v["num_latents"], d["in_channels"])
    cond = torch.randn(2, Lc, d["context_in_dim"])
    t = torch.tensor([0.5, 0.5])
    lat = torch.randn(1, v["num_latents"], v["embed_dim"])
    img = torch.randn(1, 3, cfg["cond"]["image_size"], cfg["cond"]["image_size"])
    chunk = 16000  # reference num_chunks_hy, src/config.yaml:169
    pts = torch.from_numpy(H.dense_grid_points(1.01, R)[:chunk])[None]
    n_chunks = -(-((R + 1) ** 3) // chunk)

    t0 = time.perf_counter()
    oracle_mesh = None
    try:
        oracle_mesh = omc.hy3d_mesh(grid_np, 0.0, 1.01, R)
    except (ValueError, RuntimeError):
        pass
    t_mc = time.perf_counter() - t0

    def sample(bf16, reps):
        import contextlib
        spent = [0.0]

        def tm(fn):
            best = float("inf")
            for _ in range(reps):   # min of `reps`: the first call pays allocator / thread-pool warm-up
                t1 = time.perf_counter()
                with torch.no_grad(), (torch.autocast("cpu", dtype=torch.bfloat16) if bf16 else contextlib.nullcontext()):
                    fn()
                dt_ = time.perf_counter() - t1
                spent[0] += dt_
                best = min(best, dt_)
            return best
        t_io = tm(lambda: pipe.model(x, t, cond, n_double=0, n_single=0))
        t_d = tm(lambda: pipe.model(x, t, cond, n_double=1, n_single=0)) - t_io
        t_s = tm(lambda: pipe.model(x, t, cond, n_double=0, n_single=1)) - t_io
        z = [None]
        t_vae = tm(lambda: z.__setitem__(0, pipe.vae(lat)))
        t_cond = tm(lambda: pipe.conditioner.main_image_encoder.model(img))
        t_chunk = tm(lambda: pipe.vae.geo_decoder(queries=pts, latents=z[0]))
        t_obj = (steps * (d["depth"] * t_d + d["depth_single_blocks"] * t_s + t_io) + v["num_decoder_layers"] * t_vae +
                 cfg["cond"]["num_hidden_layers"] * t_cond + n_chunks * t_chunk + t_mc)
        return t_obj, spent[0]
    t_obj, spent32 = sample(False, 3)
    t_obj16, spent16 = sample(True, 2)
    what = ("1 double + 1 single DiT block (CFG batch 2, 4442 tokens), 1 VAE layer, 1 DINOv2-g layer, 1 chunk of 16000 grid queries, "
            "C marching cubes on the full %d^3 grid; extrapolated x(%d steps x 16/32 blocks), x16 VAE, x40 DINO, x%d chunks" % (R + 1, steps, n_chunks))
    return oracle_mesh, {"value": 1.0 / t_obj, "unit": "objects/sec", "cores": torch.get_num_threads(), "kind": "port", "dtype": "fp32",
            "extrapolated": True, "seconds_per_object": t_obj, "cpu_seconds_measured": spent32 + t_mc,
            "sample": "fp32 oracle at full widths: " + what,
            "bf16": {"value": 1.0 / t_obj16, "unit": "objects/sec", "seconds_per_object": t_obj16, "cpu_seconds_measured": spent16,
                     "how": "the same modules and sample under torch.autocast('cpu', dtype=torch.bfloat16); marching cubes as in fp32"}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--model", default="full", choices=["full", "mini"])
    ap.add_argument("--inference-steps", type=int, default=50)
    ap.add_argument("--octree-resolution", type=int, default=256)
    ap.add_argument("--objects-per-launch", type=int, default=4,
                    help="crops that share the launches of the denoising loop (1 = one object at a time)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prefetch", action="store_true",
                    help="prepare every launch group's crops on the host in front of the group (round 3's behaviour)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--fp8-geo", action="store_true",
                    help="BASELINE.json configs[3] direction: the geo decoder's c_q / MLP GEMMs on fp8 (e4m3) operands; NOT the "
                         "headline configuration (the line says so in dtype / config)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the product has no CPU path)")
    # functional test of the N > 1 path on a one-GPU box: R3G_BENCH_SHARE_DEVICE=1 puts every rank on cuda:0 and switches the
    # process group to gloo (RCCL refuses two ranks on one device); never set by the driver
    share = os.environ.get("R3G_BENCH_SHARE_DEVICE") == "1"
    if share:
        local = 0
    torch.cuda.set_device(local)
    if world > 1:   # the host-side preprocess of every rank shares the box's cores
        torch.set_num_threads(max(1, (os.cpu_count() or world) // world))
    dist = None
    # the multi-rank path (crops broadcast from rank 0, meshes gathered to it, the strong block's queue) whenever the process was
    # started by torch.distributed.run -- also at WORLD_SIZE = 1 (round 5): the launch line the driver uses for N = 8 then runs
    # the same code on an RCCL group of one.  A plain `python bench.py` (the driver's N = 1 invocation) has no launcher
    # environment and stays the one-process path.
    launched = "TORCHELASTIC_RUN_ID" in os.environ or os.environ.get("R3G_BENCH_FORCE_DIST") == "1"
    if world > 1 or launched:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # (round 6) a collective timeout and a watchdog: a dead or hung peer ends the job with a message that names it inside
        # ~2 minutes instead of holding it until the launcher's limit (r3g/dist.py)
        from r3g import dist as rdist0
        if share:
            rdist0.init_process_group("gloo", rank=rank, world_size=world)
        else:
            rdist0.init_process_group("nccl", rank=rank, world_size=world, device=torch.device("cuda", local))

    from hy3dgen.shapegen import Hunyuan3DDiTFlowMatchingPipeline
    from r3g import ffi
    pipe = Hunyuan3DDiTFlowMatchingPipeline.from_pretrained("synthetic:%s:0" % a.model, device="cuda:%d" % local)
    if a.fp8_geo:
        ffi.check(ffi.lib().r3g_set_option(b"geo_fp8", 1))
    if share and world > 1:
        # several ranks on ONE device (functional runs of the N > 1 path): the ranks' query-side caches share that device's memory
        ffi.check(ffi.lib().r3g_set_option(b"geo_q_cache_gb", max(1, 64 // world)))
    cfg = pipe.cfg
    S, R = a.inference_steps, a.octree_resolution
    B = max(1, a.objects_per_launch)
    n_local = a.warmup + a.steps + B      # + one further group for the event-timed (roofline) pass
    from PIL import Image
    if dist is None:
        crops = [synthetic_crop(j) for j in range(n_local)]   # host PIL images (the stage's input format)
    else:
        # rank 0 owns the scene: all crops go out in ONE RCCL broadcast (into HBM); a rank's objects are r, r + N, ...
        from r3g import dist as rdist
        packed = [np.asarray(synthetic_crop(i)) for i in range(world * n_local)] if rank == 0 else None
        dev_crops = rdist.broadcast_crops(packed, src=0)
        crops = [Image.fromarray(dev_crops[rank + world * j].cpu().numpy(), "RGBA") for j in range(n_local)]

    def group(imgs):
        """one launch group: every object gets a generator seeded with the reference's cfg.seed (src/config.yaml:29)"""
        if len(imgs) == 1:
            return pipe(image=imgs[0], num_inference_steps=S, octree_resolution=R, num_chunks=16000,
                        generator=torch.manual_seed(1234567), output_type="raw")
        return pipe(image=list(imgs), num_inference_steps=S, octree_resolution=R, num_chunks=16000,
                    generator=[torch.Generator().manual_seed(1234567) for _ in imgs], output_type="raw")

    wd = None
    if dist is not None:
        wd = rdist.Watchdog()           # (model load and the first launches are behind us: from here a rank reports progress)

    def run(imgs, then=(), tag="run"):
        """launch group after launch group; while the GPU is in a group's denoising loop a host thread prepares the NEXT group's
        crops (recentre / resize / normalise, ~14 ms each: pipe.prefetch) -- `then` = the crops that follow `imgs`, so that a
        timed run prepares exactly as many crops inside its window as it processes (its own first group was prepared during
        the group before it, the way a persistent stage runs)"""
        out_ = []
        following = list(imgs) + list(then)
        for g0 in range(0, len(imgs), B):
            nxt = following[g0 + B:g0 + 2 * B]
            if nxt and not a.no_prefetch:
                pipe.prefetch(nxt)
            out_ += group(imgs[g0:g0 + B])
            if wd is not None:
                wd.beat("%s: launch group %d of %d done" % (tag, g0 // B + 1, -(-len(imgs) // B)))
        return out_

    def barrier():
        if dist is not None:
            rdist.barrier()
        torch.cuda.synchronize()

    run(crops[:a.warmup], then=crops[a.warmup:a.warmup + B], tag="warm-up")
    if dist is not None:
        # the mesh return's point-to-point channels (RCCL opens one per pair of ranks on first use) are opened by a one-vertex
        # gather before the clock starts, like every other first-use cost of the warm-up
        rdist.gather_meshes([(rank, torch.zeros((1, 3), dtype=torch.float32), torch.zeros((1, 3), dtype=torch.int32))], dst=0,
                            to_host=False)
    barrier()
    t0 = time.perf_counter()
    meshes = run(crops[a.warmup:a.warmup + a.steps], then=crops[a.warmup + a.steps:a.warmup + a.steps + B], tag="weak")   # EXACTLY a.steps objects
    last = meshes[-1] if meshes else None
    torch.cuda.synchronize()
    t_compute = time.perf_counter() - t0
    t_gather = 0.0
    if dist is not None:       # the meshes travel to rank 0 over RCCL (point-to-point, variable length)
        tg = time.perf_counter()
        made = [(rank + world * (a.warmup + j), m[0], m[1]) for j, m in enumerate(meshes) if m is not None]
        gathered = rdist.gather_meshes(made, dst=0, to_host=False)     # the raw meshes stay in rank 0's HBM
        if rank == 0 and len(gathered) != world * a.steps:
            raise SystemExit("gather_meshes returned %d of %d meshes" % (len(gathered), world * a.steps))
        del gathered, made
        torch.cuda.synchronize()
        t_gather = time.perf_counter() - tg
        wd.beat("weak: meshes gathered")
    tb = time.perf_counter()
    barrier()
    t_wait = time.perf_counter() - tb
    dt = time.perf_counter() - t0
    last_grid = pipe.last_grid
    per_rank_weak = None
    if dist is not None:
        tt = torch.tensor([dt], device=rdist._comm_device(), dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        # one record per rank (round 6): were < N x the objects per second load balance (compute_s spread), the mesh return
        # (gather_s) or host contention (compute_s uniformly above the one-GPU figure)?  gather_s of a rank that finished early
        # includes its wait for rank 0 to reach the gather; end_wait_s is the closing barrier.
        per_rank_weak = rdist.exchange_json({"rank": rank, "objects": len(meshes), "compute_s": round(t_compute, 4), "queue_wait_s": 0.0,
                                             "gather_s": round(t_gather, 4), "end_wait_s": round(t_wait, 4)}, name="bench_weak_ranks", dst=0)

    strong = None
    if dist is not None:
        # strong scaling: a FIXED total of 8 crops per GPU of the node (configs[2]: 8 scenes x 8 objects on 8 GPUs), claimed
        # dynamically B at a time (r3g.dist.WorkQueue), meshes gathered -- enough objects per rank for the queue to balance
        total_s = 8 * world
        q = rdist.WorkQueue(total_s, name="bench_strong", use_store=True)
        barrier()
        t1 = time.perf_counter()
        mine = []
        s_claim = s_compute = 0.0
        while True:
            tc = time.perf_counter()
            idx = q.claim_many(B)
            s_claim += time.perf_counter() - tc
            if not idx:
                break
            tc = time.perf_counter()
            ms_ = group([Image.fromarray(dev_crops[i % len(dev_crops)].cpu().numpy(), "RGBA") for i in idx])
            torch.cuda.synchronize()
            s_compute += time.perf_counter() - tc
            mine += [(i, m[0], m[1]) for i, m in zip(idx, ms_) if m is not None]
            wd.beat("strong: %d objects done" % len(mine))
        tg = time.perf_counter()
        got = rdist.gather_meshes(mine, dst=0, to_host=False)
        torch.cuda.synchronize()
        s_gather = time.perf_counter() - tg
        wd.beat("strong: meshes gathered")
        tb = time.perf_counter()
        barrier()
        s_wait = time.perf_counter() - tb
        ts = torch.tensor([time.perf_counter() - t1], device=rdist._comm_device(), dtype=torch.float64)
        dist.all_reduce(ts, op=dist.ReduceOp.MAX)
        per_rank_strong = rdist.exchange_json({"rank": rank, "objects": len(mine), "compute_s": round(s_compute, 4),
                                               "queue_wait_s": round(s_claim, 4), "gather_s": round(s_gather, 4),
                                               "end_wait_s": round(s_wait, 4)}, name="bench_strong_ranks", dst=0)
        if rank == 0:
            assert len(got) == total_s
            strong = {"objects_total": total_s, "seconds": float(ts.item()), "value": total_s / float(ts.item()),
                      "unit": "objects/sec", "assignment": "dynamic queue, %d per claim" % B, "per_rank": per_rank_strong}
        del got, mine

    out = None
    if rank == 0:
        total = world * a.steps
        out = {"metric": "objects/sec (50-step Hunyuan3D-2 DiT + 256^3 marching cubes)" if R == 256 else
                         "objects/sec (50-step Hunyuan3D-2 DiT + %d^3 marching cubes; NOT the headline metric)" % R, "value": total / dt,
               "unit": "objects/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": 1000.0 * dt / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "bf16+fp8(e4m3 operands in the geo decoder GEMMs)" if a.fp8_geo else "bf16", "data": "synthetic",
               "config": {"workload": "%s: %d synthetic 512x512 RGBA crops per GPU, Hunyuan3D-2 %s dims %s, "
                                      "%d flow-matching steps x CFG 2, %d^3 grid query + Lewiner marching cubes"
                                      % ("configs[3], shape part (the texture step is timed by tests/tex_stage_time.py, beside this line in "
                                         "profiles/)" if (a.fp8_geo and R == 512) else ("configs[1]" if (R == 256 and not a.fp8_geo) else "not a BASELINE config"),
                                         a.steps, a.model, "bf16, geo decoder c_q / MLP GEMMs on e4m3 MFMA" if a.fp8_geo else "bf16", S, R + 1),
                          "weights": "seeded synthetic", "objects_total": total, "objects_per_launch": B,
                          "parallelism": "object-parallel x%d" % world,
                          "process_group": None if dist is None else dist.get_backend()},
               "flops_per_object": flops_per_object(cfg, S, R),
               "mesh_last": None if last is None else {"V": int(last[0].shape[0]), "F": int(last[1].shape[0])}}
        # upstream's algorithmic FLOPs over the measured time.  The kernels EXECUTE about 9 % fewer FLOPs
        # (CFG de-duplication), so the hardware's own utilisation is lower than this figure: see "..._executed" below.
        out["mfma_utilisation_end_to_end"] = out["flops_per_object"] * out["value"] / world / (PEAK_BF16_TFLOPS * 1e12)
        if per_rank_weak is not None:
            out["per_rank"] = per_rank_weak
        if strong is not None:
            out["strong"] = strong
        # launch groups whose fp16 residual stream overflowed and that ran again on the fp32 stream (twice their time), of all
        # launch groups this process ran (r3g_get_counter)
        out["dit_f16_fallbacks"] = ffi.counter("dit_f16_fallbacks")
        out["dit_groups"] = ffi.counter("dit_groups")

    if rank == 0 and not a.no_roofline:
        L = ffi.lib()
        # one further object with every launch bracketed by HIP events on its stream.  The optional second-stream
        # overlap (a GEMM beside the attention kernel; off by default) stays off for this object so that a launch
        # duration is the kernel's own time, not the time it shared the GPU with another kernel.
        overlap_on = "overlap_mlp=1" in os.environ.get("R3G_OPTIONS", "")
        ffi.check(L.r3g_set_option(b"overlap_mlp", 0))
        ffi.check(L.r3g_prof_enable(1))
        prof_meshes = run(crops[a.warmup + a.steps:a.warmup + a.steps + B])       # one further launch group of B objects
        torch.cuda.synchronize()
        ffi.check(L.r3g_set_option(b"overlap_mlp", 1 if overlap_on else 0))
        n = len(FAMILIES)
        cnt, ms, work = (ctypes.c_int64 * n)(), (ctypes.c_double * n)(), (ctypes.c_double * n)()
        ffi.check(L.r3g_prof_read(cnt, ms, work, n))
        alg = (ctypes.c_double * n)()
        ffi.check(L.r3g_prof_read_bytes(alg, n))
        ffi.check(L.r3g_prof_enable(0))
        # per OBJECT: the event-timed pass ran one group of B objects
        fam = {FAMILIES[i]: {"launches": int(cnt[i]), "ms": float(ms[i]) / B, "work": float(work[i]) / B} for i in range(n)}
        dom = max(("gemm", "attention"), key=lambda k: fam[k]["ms"])
        ach = fam[dom]["work"] / (fam[dom]["ms"] * 1e-3) / 1e12 if fam[dom]["ms"] > 0 else 0.0
        executed = fam["gemm"]["work"] + fam["attention"]["work"]      # FLOPs the MFMA kernels actually issued
        out["flops_per_object_executed"] = executed
        out["mfma_utilisation_executed"] = executed * out["value"] / world / (PEAK_BF16_TFLOPS * 1e12)
        tr = recorded_traffic(dom)
        out["roofline"] = {"kernel": dom, "bound": "mfma", "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                           "frac": ach / PEAK_BF16_TFLOPS,
                           "traffic": None if tr is None else tr.get("bytes_per_launch"),
                           "traffic_source": None if tr is None else
                           "profiles/traffic.json (%s, commit %s)" % (tr.get("measured"), tr.get("commit")),
                           # True: the PMC record was measured on a different build of libr3g.so than this run loaded
                           "traffic_stale": None if tr is None else bool(tr.get("stale")),
                           "algorithmic_bytes_per_launch": (float(alg[FAMILIES.index(dom)]) / max(1, fam[dom]["launches"])
                                                            if dom == "gemm" else None),
                           "launches": fam[dom]["launches"],
                           "avg_launch_us": 1000.0 * B * fam[dom]["ms"] / max(1, fam[dom]["launches"]),
                           "note": "per-launch HIP events on one extra launch group of %d objects (one stream: kernels run "
                                   "alone); families_ms_per_object = group time / %d" % (B, B),
                           "families_ms_per_object": {k: round(v["ms"], 3) for k, v in fam.items()},
                           "attention_tflops": (fam["attention"]["work"] / (fam["attention"]["ms"] * 1e-3) / 1e12
                                                if fam["attention"]["ms"] > 0 else 0.0),
                           "mc_classify_gbps": (fam["mc_classify"]["work"] / (fam["mc_classify"]["ms"] * 1e-3) / 1e9
                                                if fam["mc_classify"]["ms"] > 0 else 0.0)}
        # the HBM-bound kernel family of the path: marching cubes on the resident grid (one read of the grid + one
        # write of the mesh); V, F of the last object of the event-timed group
        mc_ms = fam["mc_classify"]["ms"] + fam["mc_other"]["ms"]
        prof_meshes = [m for m in prof_meshes if m is not None]
        if prof_meshes and mc_ms > 0:
            mc_bytes = 4.0 * (R + 1) ** 3 + 12.0 * sum(int(m[0].shape[0]) + int(m[1].shape[0]) for m in prof_meshes) / len(prof_meshes)
            out["roofline_mc"] = {"kernel": "marching cubes (classify + scan + vertices + faces)", "bound": "hbm",
                                  "achieved": mc_bytes / (mc_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                                  "frac": mc_bytes / (mc_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS, "ms": mc_ms,
                                  "algorithmic_bytes": mc_bytes,
                                  "note": "synthetic weights give a noise-like field: V and F are ~5x a real object's"}
        # the same kernels on an object-like field (a union of smooth blobs, ~1e5 vertices as real objects give):
        # the synthetic weights' noise-like field has ~8x the surface and is dominated by the per-cell emission kernels
        try:
            from r3g import mc as rmc
            ax = torch.linspace(-1.0, 1.0, R + 1, device="cuda")
            X, Y, Z = torch.meshgrid(ax, ax, ax, indexing="ij")
            rng = np.random.default_rng(7)
            fld = torch.full_like(X, -1.0)
            for _ in range(5):
                cx, cy, cz = rng.uniform(-0.35, 0.35, 3)
                fld = torch.maximum(fld, float(rng.uniform(0.3, 0.5)) - torch.sqrt((X - cx) ** 2 + (Y - cy) ** 2 + (Z - cz) ** 2))
            del X, Y, Z
            fld = fld.contiguous()
            rmc.extract_mesh(fld, 0.0, 1.01, R)
            torch.cuda.synchronize()
            ffi.check(L.r3g_prof_enable(1))
            sv, sf = rmc.extract_mesh(fld, 0.0, 1.01, R)
            torch.cuda.synchronize()
            ffi.check(L.r3g_prof_read(cnt, ms, work, n))
            ffi.check(L.r3g_prof_enable(0))
            sm_ms = float(ms[FAMILIES.index("mc_classify")]) + float(ms[FAMILIES.index("mc_other")])
            sm_bytes = 4.0 * (R + 1) ** 3 + 12.0 * (int(sv.shape[0]) + int(sf.shape[0]))
            if sm_ms > 0 and "roofline_mc" in out:
                out["roofline_mc"]["object_like_field"] = {
                    "V": int(sv.shape[0]), "F": int(sf.shape[0]), "ms": sm_ms, "achieved": sm_bytes / (sm_ms * 1e-3) / 1e9,
                    "frac": sm_bytes / (sm_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS,
                    "note": "same kernels, smooth union-of-blobs field on the same %d^3 grid (not part of the timed objects)" % (R + 1)}
            del fld, sv, sf
        except Exception as e:      # reporting only: never fails the bench line
            out.setdefault("roofline_mc", {})["object_like_field"] = {"error": str(e)}
    bad = False
    if hasattr(pipe, "close_prefetch"):
        pipe.close_prefetch()       # the host-preparation thread ends here
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        oracle_mesh, out["cpu_baseline"] = cpu_baseline(cfg, S, R, last_grid.cpu().numpy())
        # the C oracle's mesh of the last TIMED object's grid against the mesh the timed region produced
        if oracle_mesh is None or last is None:
            out["mc_parity"] = "no surface"
        else:
            ov, of = oracle_mesh
            gv, gf = last[0].cpu().numpy(), last[1].cpu().numpy()
            same = (gv.shape == ov.shape and gf.shape == of.shape and np.array_equal(gf.astype(np.int64), of.astype(np.int64))
                    and np.array_equal(gv.astype(np.float32).view(np.uint32), ov.astype(np.float32).view(np.uint32)))
            out["mc_parity"] = "exact" if same else "MISMATCH"
            bad = not same
    if rank == 0:
        # host side of a crop: mean milliseconds per preparation, and how many of the run's crops were picked up from the
        # host thread instead of being prepared in front of their launch group (round 5: all but the very first group's)
        out["host_prepare_ms_per_object"] = 1000.0 * pipe.timings.get("host_prepare_s", 0.0) / max(1, pipe.timings.get("host_prepare_n", n_local))
        out["host_prepared_ahead"] = "%d of %d" % (pipe.timings.get("prefetch_hits", 0), pipe.timings.get("host_prepare_n", 0))
        print(json.dumps(out))
    if bad:
        raise SystemExit("bench.py: the timed object's mesh differs from the marching-cubes oracle's mesh of the same grid")
    if dist is not None:
        wd.done()
        rdist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
