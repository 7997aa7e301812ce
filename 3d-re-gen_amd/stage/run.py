#!/usr/bin/env python3
"""Drop-in for the reference stage script src/2d_to_3d_models/run.py (pipeline name "Hunyuan_2d_to_3d").

Same CLI (`--config <yaml>`), same YAML keys (src/config.yaml: mini, num_inf_steps_hy, octree_resolution_hy,
num_chunks_hy, seed, remesh, remesh_target_num_faces, input_folder_hy, output_folder_hy, use_banana,
prepped_for_hunyuan, jobs_per_gpu, use_all_available_cuda), same filesystem contract (reference :160-173, :99-102):
every *.png/*.jpg/*.jpeg of the input folder except names containing wall/walls/room/ceiling/floor; the output
folder is created and emptied; one `<out>/<stem>/<stem>.glb` per image; FileNotFoundError without images.

What differs, by design (MI355X-first):
  * one persistent process per GPU (torch.distributed over RCCL; the reference spawns one process per IMAGE and
    reloads both pipelines each time, :119-130) -- the model is loaded once per rank;
  * with more than one GPU, rank 0 decodes the crops and broadcasts the packed batch over RCCL, ranks claim object
    indices dynamically from a shared counter (no static i % num_devices split, reference :188-191), and the cleaned
    meshes are gathered to rank 0, which writes the GLBs (r3g/dist.py); files are the SORTED list (the reference's
    os.listdir order is filesystem dependent); the result is byte-identical to the one-process run;
  * per-object failures are collected into a status list and printed (the reference's pool path swallows them,
    :135-136); the exit code follows the reference: 0 when the stage ran, non-zero only on setup errors
    (no images / no weights), or when the sequential path hits an exception (as in the reference, :212-213);
  * weights: private key `r3g_weights` (or env R3G_WEIGHTS) = local snapshot directory or 'synthetic:<full|mini>';
    without it the HF cache is consulted offline, as there is no network on the target machines.
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
if PKG not in sys.path:
    sys.path.insert(0, PKG)

IMAGE_EXTENSIONS = (".png", ".jpg", ".jpeg")
DONT_MESH = ["wall", "walls", "room", "ceiling", "floor"]


def load_config(path):
    """src/utils/global_utils.py:464-476"""
    import yaml
    if not os.path.exists(path):
        raise FileNotFoundError("Config file not found: %s" % path)
    with open(path, "r") as f:
        return yaml.safe_load(f)


def clear_output_directory(output_dir):
    """src/utils/global_utils.py:443-461"""
    if not os.path.exists(output_dir):
        raise FileNotFoundError("Output directory does not exist: %s" % output_dir)
    for item in os.listdir(output_dir):
        p = os.path.join(output_dir, item)
        if os.path.isdir(p):
            shutil.rmtree(p)
        else:
            os.remove(p)


def list_images(input_folder):
    names = sorted(f for f in os.listdir(input_folder)
                   if f.lower().endswith(IMAGE_EXTENSIONS) and not any(x in f.lower() for x in DONT_MESH))
    if not names:
        raise FileNotFoundError("No images found in the input folder '%s'." % input_folder)
    return [os.path.join(input_folder, f) for f in names]


def select_model(config):
    """reference :146-157 -- model ids / from_pretrained kwargs, plus the local weights override"""
    models = {"full": {"id": "tencent/Hunyuan3D-2", "args": {}},
              "mini": {"id": "tencent/Hunyuan3D-2mini", "args": {"subfolder": "hunyuan3d-dit-v2-mini", "variant": "fp16"}}}
    key = "mini" if config.get("mini", True) else "full"
    return key, models[key]


def resolve_weights(config, key, model):
    w = config.get("r3g_weights") or os.environ.get("R3G_WEIGHTS")
    if w:
        return w.replace("{model}", key)
    try:
        from huggingface_hub import snapshot_download
        return snapshot_download(repo_id=model["id"], local_files_only=True)
    except Exception as e:
        raise FileNotFoundError("no local weights for %s: set `r3g_weights:` in the config (a snapshot directory or "
                                "'synthetic:%s') -- there is no network here (%s)" % (model["id"], key, e))


def default_factory(config, device):
    """-> (pipeline_shapegen, pipeline_texgen, cleaners) using the MI355X hy3dgen mirror"""
    from hy3dgen.shapegen import (DegenerateFaceRemover, FaceReducer, FloaterRemover,
                                  Hunyuan3DDiTFlowMatchingPipeline)
    from hy3dgen.texgen import Hunyuan3DPaintPipeline
    key, model = select_model(config)
    print("Using '%s' shape generator: %s" % (key, model["id"]))
    path = resolve_weights(config, key, model)
    shapegen = Hunyuan3DDiTFlowMatchingPipeline.from_pretrained(path, device=device, **model["args"])
    # the texture models go on this rank's GPU; a snapshot whose texture folders cannot be read (e.g. no
    # prompt_embeds_empty.safetensors beside a stock checkpoint: INTEGRATION.md) must not take the shape stage down with it
    # private key `r3g_texture_weights`: the folder that holds upstream's two texture sub-folders (hunyuan3d-delight-v2-0,
    # hunyuan3d-paint-v2-0) when it is not the shape model's snapshot -- e.g. shape weights "synthetic:..." beside real or
    # test-written texture checkpoints
    tex_path = config.get("r3g_texture_weights") or path
    # Round 5: a texture sub-folder that EXISTS and cannot be loaded fails the stage, as upstream does -- a service must
    # not silently write untextured or single-view GLBs.  `r3g_require_textures: false` in the YAML is the explicit opt-in to
    # best-effort loading (the problems are then printed AND carried in the stage's JSON report as `texture_load_problems`).
    tex_dirs = [os.path.join(str(tex_path), sub) for sub in (Hunyuan3DPaintPipeline.DELIGHT_SUBFOLDER,
                                                              Hunyuan3DPaintPipeline.MULTIVIEW_SUBFOLDER)]
    have_tex = any(os.path.isdir(d) for d in tex_dirs)
    strict = bool(config.get("r3g_require_textures", have_tex))
    texgen = Hunyuan3DPaintPipeline.from_pretrained(tex_path, device=device, strict=strict)
    for problem in getattr(texgen, "load_problems", []):
        print("[WARN] texture model not loaded, continuing without it -- %s" % problem, file=sys.stderr)
    LOAD_PROBLEMS[:] = list(getattr(texgen, "load_problems", []))
    return shapegen, texgen, [FloaterRemover(), DegenerateFaceRemover(), FaceReducer()]


def clean_and_validate_mesh(mesh, min_faces=10, target_face_count=None):
    """reference clean_and_validate_trimesh :24-64 on the Trimesh-like r3g.mesh.Mesh"""
    import numpy as np
    if mesh is None or mesh.is_empty:
        raise ValueError("Input is not a valid or is an empty trimesh object.")
    ok = np.all(np.isfinite(mesh.vertices), axis=1)
    if not ok.all():
        print("[WARN] Found %d invalid (NaN/Inf) vertices. Cleaning..." % int((~ok).sum()))
        mesh.update_vertices(ok)
    if target_face_count is not None and len(mesh.faces) > target_face_count:
        print("Simplifying mesh from %d to %d faces..." % (len(mesh.faces), target_face_count))
        mesh = mesh.simplify_quadric_decimation(face_count=target_face_count)
        print("Simplified mesh has %d faces." % len(mesh.faces))
    if mesh.is_empty or len(mesh.faces) < min_faces:
        raise ValueError("Mesh is empty or has fewer than %d faces after cleaning/simplification." % min_faces)
    mesh.process(validate=True)
    mesh.remove_unreferenced_vertices()
    mesh.update_faces(mesh.nondegenerate_faces())
    if mesh.is_empty or len(mesh.faces) < min_faces:
        raise ValueError("Mesh became empty after final processing.")
    return mesh


def objects_per_launch(config):
    """private key `r3g_objects_per_launch` (default 4): how many crops share the launches of the denoising loop.  The DiT's
    GEMMs have 7.5 k rows per object; from two objects on every layer has enough rows for 256x256 tiles on all 256 CUs
    (measured: 1 663 / 1 499 / 1 476 ms per object at 1 / 2 / 4 objects per launch).  Results do not depend on it
    (bit-identical per object)."""
    return max(1, int(config.get("r3g_objects_per_launch", 4)))


def shape_meshes(images, shapegen, config):
    """reference :77-84 for a group of images: the raw marching-cubes meshes, in order (None where extraction failed).
    Every object gets a generator seeded with cfg.seed, exactly as the reference seeds each of its calls (:82)."""
    import torch
    kw = dict(num_inference_steps=config.get("num_inf_steps_hy", 100), octree_resolution=config.get("octree_resolution_hy", 380),
              num_chunks=config.get("num_chunks_hy", 20000), output_type="trimesh")
    seed = config.get("seed", 12345)
    if len(images) > 1 and getattr(shapegen, "accepts_image_list", False):
        return list(shapegen(image=list(images), generator=[torch.Generator().manual_seed(seed) for _ in images], **kw))
    return [shapegen(image=im, generator=torch.manual_seed(seed), **kw)[0] for im in images]


LOAD_PROBLEMS = []      # texture checkpoints that were present and could not be loaded (best-effort mode only): goes into the report

# seconds of the two phases behind the shape model, per finished object of this process (the stage's report carries their means:
# what a `-p 3` user waits for beyond the metric of bench.py -- SURVEY 8d excludes cleaners and texture from objects/sec)
PHASE_SECONDS = {"cleaners": [], "texture": []}


def finish_mesh(mesh, image, texgen, cleaners, config):
    """reference process_image :86-97: optional remesh, the three cleaners, the texture stage"""
    if mesh is None:
        raise RuntimeError("surface extraction produced no mesh")
    if config.get("remesh", False):
        mesh = clean_and_validate_mesh(mesh, target_face_count=config.get("remesh_target_num_faces", 30000))
    print("Initial mesh has %d vertices and %d faces." % (mesh.n_vertices, mesh.n_faces))
    t0 = time.time()
    for cleaner in cleaners:
        mesh = cleaner(mesh)
    print("Cleaned mesh has %d vertices and %d faces." % (mesh.n_vertices, mesh.n_faces))
    t1 = time.time()
    mesh = texgen(mesh, image=image)
    PHASE_SECONDS["cleaners"].append(t1 - t0)
    PHASE_SECONDS["texture"].append(time.time() - t1)
    return mesh


def generate_mesh(image, base, shapegen, texgen, cleaners, config):
    """reference process_image :69-97 without the file I/O around it: RGBA image -> cleaned (and textured) mesh"""
    print("Processing %s..." % base)
    return finish_mesh(shape_meshes([image], shapegen, config)[0], image, texgen, cleaners, config)


def generate_group(images, bases, shapegen, texgen, cleaners, config, isolate):
    """A group of crops through the shape pipeline together, then one by one through cleaners + texture stage.
    -> [(mesh or None, error or None, seconds)] in order.  isolate: an exception does not leave the group (the object is
    reported as failed; a failure of the shared shape launch is retried object by object so that one bad crop cannot
    take its neighbours down); otherwise the first exception propagates, as in the reference's sequential path."""
    t0 = time.time()
    for b in bases:
        print("Processing %s..." % b)
    try:
        raw = shape_meshes(images, shapegen, config)
        errs = [None] * len(images)
    except Exception as e:
        if not isolate:
            raise
        raw, errs = [], []
        for im in images:
            try:
                raw.append(shape_meshes([im], shapegen, config)[0] if len(images) > 1 else None)
                errs.append(None if len(images) > 1 else e)
            except Exception as e1:
                raw.append(None)
                errs.append(e1)
    t_shape = (time.time() - t0) / max(1, len(images))
    out = []
    for im, m, err in zip(images, raw, errs):
        t1 = time.time()
        if err is not None:
            out.append((None, err, t_shape))
            continue
        try:
            out.append((finish_mesh(m, im, texgen, cleaners, config), None, t_shape + time.time() - t1))
        except Exception as e:
            if not isolate:
                raise
            out.append((None, e, t_shape + time.time() - t1))
    return out


def export_mesh(mesh, base, output_dir):
    """reference :99-102: <out>/<stem>/<stem>.glb"""
    out_dir = os.path.join(output_dir, base)
    os.makedirs(out_dir, exist_ok=True)
    out_path = os.path.join(out_dir, base + ".glb")
    mesh.export(out_path)
    return out_path


def process_image(image_path, shapegen, texgen, cleaners, output_dir, config):
    """reference process_image :67-105"""
    from PIL import Image
    image = Image.open(image_path).convert("RGBA")
    base = os.path.splitext(os.path.basename(image_path))[0]
    t0 = time.time()
    mesh = generate_mesh(image, base, shapegen, texgen, cleaners, config)
    out_path = export_mesh(mesh, base, output_dir)
    print("Saved %s to %s in %.2f seconds." % (base, out_path, time.time() - t0))
    return out_path


def partition(n_items, rank, world):
    """static round-robin over the sorted list (reference: i % num_devices, :188-191); the distributed path hands
    indices out dynamically instead (r3g.dist.WorkQueue) -- kept for tools that want a fixed split"""
    return list(range(rank, n_items, world))


def _stem(path):
    return os.path.splitext(os.path.basename(path))[0]


def run_rank(config, image_paths, output_folder, rank, world, factory, swallow_errors):
    """Sequential path (one process): load the model once, process the images in groups of `r3g_objects_per_launch`,
    return [(index, path, status, seconds)]."""
    import torch
    from PIL import Image
    device = "cuda:%d" % int(os.environ.get("LOCAL_RANK", "0")) if torch.cuda.is_available() else "cpu"
    shapegen, texgen, cleaners = factory(config, device)
    results = []
    todo = partition(len(image_paths), rank, world)
    B = objects_per_launch(config)
    opened = {}
    # GLB encoding (PNG deflate of the texture, 50-80 ms per object) runs on a host thread while the next object is on the GPU;
    # the mesh's arrays are brought to the host first, on this thread
    import concurrent.futures
    writer = concurrent.futures.ThreadPoolExecutor(max_workers=1, thread_name_prefix="r3g-glb-writer")
    pending = []

    def open_group(g0):
        if g0 not in opened:
            opened[g0] = [Image.open(image_paths[i]).convert("RGBA") for i in todo[g0:g0 + B]]
        return opened[g0]
    for g0 in range(0, len(todo), B):
        idx = todo[g0:g0 + B]
        images = open_group(g0)
        opened.pop(g0)
        if g0 + B < len(todo) and hasattr(shapegen, "prefetch"):
            # the next group's crops are decoded and prepared on a host thread while this group is on the GPU
            shapegen.prefetch(open_group(g0 + B))
        bases = [_stem(image_paths[i]) for i in idx]
        for i, base, (mesh, err, secs) in zip(idx, bases, generate_group(images, bases, shapegen, texgen, cleaners, config,
                                                                          isolate=swallow_errors)):
            if err is None:
                try:
                    _ = (mesh.vertices, mesh.faces, getattr(mesh, "uv", None), getattr(mesh, "texture", None))   # device -> host
                    pending.append((i, base, secs, writer.submit(export_mesh, mesh, base, output_folder)))
                except Exception as e:        # noqa: BLE001
                    if not swallow_errors:
                        raise
                    print("ERROR in worker for '%s' on rank %d: %s" % (os.path.basename(image_paths[i]), rank, e))
                    results.append((i, image_paths[i], "error: %s" % e, secs))
            else:
                print("ERROR in worker for '%s' on rank %d: %s" % (os.path.basename(image_paths[i]), rank, err))
                results.append((i, image_paths[i], "error: %s" % err, secs))
        done_now, pending = [p for p in pending if p[3].done()], [p for p in pending if not p[3].done()]
        for item in done_now:
            _collect_export(item, results, image_paths, rank, swallow_errors)
    for item in pending:
        _collect_export(item, results, image_paths, rank, swallow_errors)
    writer.shutdown()
    if hasattr(shapegen, "close_prefetch"):
        shapegen.close_prefetch()
    results.sort(key=lambda r: r[0])
    return results, texture_state(texgen)


def _collect_export(item, results, image_paths, rank, swallow_errors):
    i, base, secs, fut = item
    try:
        out_path = fut.result()
        print("Saved %s to %s in %.2f seconds." % (base, out_path, secs))
        results.append((i, image_paths[i], "ok", secs))
    except Exception as e:        # noqa: BLE001
        if not swallow_errors:
            raise
        print("ERROR in worker for '%s' on rank %d: %s" % (os.path.basename(image_paths[i]), rank, e))
        results.append((i, image_paths[i], "error: %s" % e, secs))


def run_distributed(config, input_folder, output_folder, rank, world, factory):
    """One persistent rank per GPU.  Rank 0 decodes the crops and broadcasts the packed batch (RCCL: into every rank's
    HBM); ranks claim object indices from a shared counter as they become free (`r3g_objects_per_launch` at a time), keep
    their meshes, and rank 0 gathers and writes them.  A rank whose setup or loop dies (model load, out of memory, ...)
    still joins every collective: its finished meshes are delivered, the objects it had claimed are reported as failed,
    and the stage's exit code says so.  Returns (results, textured, failed_ranks) on rank 0, (None, None, failed_ranks)
    elsewhere."""
    import numpy as np
    import torch
    from PIL import Image
    from r3g import dist as rdist
    from r3g.mesh import Mesh
    device = "cuda:%d" % int(os.environ.get("LOCAL_RANK", "0")) if torch.cuda.is_available() else "cpu"
    crops = None
    image_paths = None
    if rank == 0:
        image_paths = list_images(input_folder)
        crops = [np.asarray(Image.open(p).convert("RGBA")) for p in image_paths]
    image_paths = rdist.share_json(image_paths, src=0, name="r3g_stage_names")    # the file names only
    crops = rdist.broadcast_crops(crops, src=0)                                     # the pixels travel as one tensor
    queue = rdist.WorkQueue(len(image_paths), name="r3g_stage_objects")
    # r3g_stream_outputs: every rank writes `<out>/<stem>/<stem>.glb` the moment the object is done (a consumer such as the
    # scene-reconstruction stage can start per object) instead of returning the mesh to rank 0; the files
    # are the same bytes either way (the writer is deterministic)
    stream_out = bool(config.get("r3g_stream_outputs", False))
    B = objects_per_launch(config)
    mine, status = [], []
    rank_error = None
    shapegen = texgen = None
    claimed = []
    # a rank that dies or hangs ends the stage with a message that names it (r3g/dist.py Watchdog; `r3g_watchdog_s`: seconds
    # without progress -- model load included -- after which a rank counts as lost, default 300; 0 = no watchdog)
    wd_limit = float(config.get("r3g_watchdog_s", 300))
    wd = rdist.Watchdog(limit_s=wd_limit) if wd_limit > 0 else None
    try:
        if wd is not None:
            wd.beat("loading the models")
        shapegen, texgen, cleaners = factory(config, device)
        if wd is not None:
            wd.beat("models loaded")

        def images_of(idx):
            return [Image.fromarray(crops[i].cpu().numpy(), "RGBA") for i in idx]
        # Claims are guided (a full launch group while the list is long, smaller ones when it runs short: every rank gets
        # work) and, while every rank can still get a full group, one group AHEAD: while a group is on the GPU the next one
        # is already claimed and its crops are prepared on the pipeline's host thread (prefetch), as in the one-process path;
        # near the end of the list a rank claims only when it is free, so that nobody sits on objects an idle GPU could take.
        # `claimed` holds everything this rank has taken and not reported yet -- a rank-level failure reports all of it.
        claimed = queue.claim_guided(B, world)
        ahead_images = None
        while claimed:
            current = list(claimed)
            images = ahead_images if ahead_images is not None else images_of(current)
            nxt = queue.claim_guided(B, world) if queue.remaining() >= B * world else []
            claimed = current + nxt
            ahead_images = None
            if nxt and hasattr(shapegen, "prefetch"):
                ahead_images = images_of(nxt)
                shapegen.prefetch(ahead_images)
            bases = [_stem(image_paths[i]) for i in current]
            # one object failing must not fail the stage (reference :135-136 swallows it silently)
            for i, base, (mesh, err, secs) in zip(current, bases, generate_group(images, bases, shapegen, texgen, cleaners,
                                                                                  config, isolate=True)):
                if err is None:
                    if stream_out:
                        export_mesh(mesh, base, output_folder)
                    else:
                        mine.append((i, mesh))
                    status.append((i, image_paths[i], "ok", secs, rank))
                else:
                    print("ERROR in worker for '%s' on rank %d: %s" % (os.path.basename(image_paths[i]), rank, err))
                    status.append((i, image_paths[i], "error: %s" % err, secs, rank))
            if wd is not None:
                wd.beat("%d objects done, last '%s'" % (len(status), bases[-1]))
            claimed = nxt if nxt else queue.claim_guided(B, world)
    except Exception as e:      # rank-level failure: keep what is finished, report what was in flight, stay collective
        rank_error = e
        print("ERROR on rank %d (rank-level, outside the per-object handling): %r" % (rank, e), file=sys.stderr)
        done = {s[0] for s in status}
        for i in claimed:
            if i not in done:
                status.append((i, image_paths[i], "error: rank %d failed: %s" % (rank, e), 0.0, rank))
    if shapegen is not None and hasattr(shapegen, "close_prefetch"):
        shapegen.close_prefetch()
    if wd is not None:
        wd.done()           # (a rank that waits for the others in the gather below is finished, not hung)
    ok_flags = rdist.all_ok(rank_error is None)
    failed_ranks = [r for r, ok in enumerate(ok_flags) if not ok]
    local = []
    for i, mesh in mine:
        if getattr(mesh, "_dv", None) is not None:
            local.append((i, mesh._dv, mesh._df))          # still in HBM: goes over RCCL from there
        else:
            local.append((i, np.asarray(mesh.vertices, np.float32), np.asarray(mesh.faces, np.int32),
                          getattr(mesh, "uv", None), getattr(mesh, "texture", None)))     # textured: uv + PNG source pixels too
    gathered = rdist.gather_meshes(local, dst=0)
    all_status = rdist.exchange_json(status, name="r3g_stage_status", dst=0)
    if rank != 0:
        return None, None, failed_ranks
    for i in sorted(gathered):
        g = gathered[i]
        mesh = Mesh(g[0], g[1]) if len(g) == 2 else Mesh(g[0], g[1], uv=g[2], texture=g[3])
        export_mesh(mesh, _stem(image_paths[i]), output_folder)
    results = sorted(tuple(r) for part in all_status for r in part)
    # objects nobody reported (claimed by a rank that died before it could say so) are failures, not silence
    seen = {r[0] for r in results}
    for i in range(len(image_paths)):
        if i not in seen:
            results.append((i, image_paths[i], "error: not processed (ranks %s failed)" % failed_ranks, 0.0, -1))
    return sorted(results), texture_state(texgen) if texgen is not None else (True, None), failed_ranks


def main(argv=None, factory=default_factory):
    ap = argparse.ArgumentParser(description="Run 2D to 3D model generation (MI355X-native).")
    ap.add_argument("--config", default="../src/config.yaml", type=str, help="Path to the configuration file.")
    args = ap.parse_args(argv)
    config = load_config(args.config)
    input_folder = config["input_folder_hy"]
    if config["use_banana"]:
        input_folder = config["prepped_for_hunyuan"]
    output_folder = config["output_folder_hy"]

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    import torch
    if world == 1:
        os.makedirs(output_folder, exist_ok=True)
        clear_output_directory(output_folder)
        image_paths = list_images(input_folder)
        n_dev = torch.cuda.device_count()
        slots = n_dev * max(1, int(config.get("jobs_per_gpu", 1)))
        if n_dev > 1 and len(image_paths) > 1 and slots > 1 and factory is default_factory:
            # re-launch as one persistent rank per GPU (torch.distributed over RCCL)
            n = min(n_dev, len(image_paths))
            print("Found %d GPU(s): launching %d persistent rank(s)." % (n_dev, n))
            env = dict(os.environ, R3G_STAGE_PREPARED="1")
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
                   "--master-addr", "127.0.0.1", "--master-port", str(29500 + os.getpid() % 2000),
                   os.path.abspath(__file__), "--config", args.config]
            return subprocess.run(cmd, env=env, check=True).returncode
        print("Running sequentially (%s)." % ("no GPU found" if n_dev == 0 else "only 1 slot or only 1 image"))
        results, textured = run_rank(config, image_paths, output_folder, 0, 1, factory, swallow_errors=False)
        return finish(report(results, textured), config)

    # ---- distributed: rank 0 prepares the output folder and scatters the crops, everybody works, rank 0 gathers
    import torch.distributed as dist
    backend = "nccl" if torch.cuda.is_available() else "gloo"
    if torch.cuda.is_available():
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    from r3g import dist as rdist
    # (collective timeout, R3G_DIST_TIMEOUT_S: a collective a dead peer never joins raises instead of waiting for ever)
    rdist.init_process_group(backend, device=torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0"))) if torch.cuda.is_available() else None)
    rc = 0
    try:
        if rank == 0 and not os.environ.get("R3G_STAGE_PREPARED"):
            os.makedirs(output_folder, exist_ok=True)
            clear_output_directory(output_folder)
        rdist.barrier()
        results, textured, failed_ranks = run_distributed(config, input_folder, output_folder, rank, world, factory)
        if rank == 0:
            rc = finish(report(results, textured), config)
            print("All parallel tasks completed.")
        if failed_ranks:      # a rank-level failure (not a per-object one) is a setup error: non-zero, on every rank
            print("[r3g] rank(s) %s failed outside the per-object handling" % failed_ranks, file=sys.stderr)
            rc = rc or 4
        rdist.barrier()       # rank 0 serves the side store and writes the files: nobody tears the group down before it is done
    finally:
        rdist.reset()
        dist.destroy_process_group()
    return rc


def texture_state(texgen):
    """(textured, where the colours come from) of a texgen pipeline object"""
    return bool(getattr(texgen, "implemented", True)), getattr(texgen, "source", None)


def report(results, textured=True):
    source = None
    if isinstance(textured, tuple):
        textured, source = textured
    ok = sum(1 for r in results if r[2] == "ok")
    rep = {"stage": "Hunyuan_2d_to_3d", "objects": len(results), "ok": ok,
           "failed": [os.path.basename(r[1]) for r in results if r[2] != "ok"],
           "seconds": [round(r[3], 3) for r in results],
           "mean_seconds_cleaners": round(sum(PHASE_SECONDS["cleaners"]) / max(1, len(PHASE_SECONDS["cleaners"])), 3),
           "mean_seconds_texture": round(sum(PHASE_SECONDS["texture"]) / max(1, len(PHASE_SECONDS["texture"])), 3),
           "textured": bool(textured)}
    if source:
        rep["texture_source"] = source
    if LOAD_PROBLEMS:
        rep["texture_load_problems"] = list(LOAD_PROBLEMS)
    if results and len(results[0]) > 4:
        rep["rank_of_object"] = [r[4] for r in results]
    try:
        # launch groups of THIS process (rank 0 of a distributed run) whose fp16 residual stream overflowed and that ran again on
        # the fp32 stream (r3g_get_counter): such a group takes twice its time -- a slow stage says why
        from r3g import ffi
        rep["dit_f16_fallbacks"] = ffi.counter("dit_f16_fallbacks")
        rep["dit_groups"] = ffi.counter("dit_groups")
    except Exception:       # (the CPU-only API-contract tests run this script without the library)
        pass
    print(json.dumps(rep))
    return rep


def finish(rep, config):
    """exit code of the stage.  GLBs without a base-colour texture (a texgen object with `implemented = False`) are
    reported (`"textured": false`) and warned about; `r3g_require_textures: true` in the config turns that into a failure
    so that a pipeline that needs textures does not silently run on without them.  When the texture does not come from
    the multiview diffusion upstream uses (this path has no such model), `texture_source` says where it comes from."""
    if not rep["textured"]:
        msg = ("[r3g] WARNING: no texture stage on this run: %d GLB(s) written WITHOUT baseColorTexture (geometry only)"
               % rep["ok"])
        print(msg, file=sys.stderr)
        if config.get("r3g_require_textures", False):
            print("[r3g] r3g_require_textures is set: failing the stage", file=sys.stderr)
            return 3
    elif rep.get("texture_source"):
        print("[r3g] textures: %s" % rep["texture_source"], file=sys.stderr)
    return 0


if __name__ == "__main__":
    sys.exit(main())
