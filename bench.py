#!/usr/bin/env python3
"""bench.py — Msamples/s of the path_mis hot path on MI355X.

A "step" is one full render pass of the workload: every pixel, every sample, film accumulation included, with
the scene resident in HBM.  At N > 1 GPUs (one process per GPU, launched by torch.distributed.run) every rank
renders its own spp slice of the same image (weak scaling: per-GPU work is fixed) and the fp32 film is summed
over ranks with one RCCL all-reduce per step, inside the timed region.

Workloads (BASELINE.json configs):
  table  scenes/table/table_path_mis_microfacet.xml 800x600, 256 spp   (configs[1]: the single-GPU configuration; default)
  cbox   scenes/cbox/cbox_path_mis.xml at 512x512, 64 spp      (configs[0], the reference's CPU-runnable case)
  dragon scenes/proj/dragon-diffuse.xml 800x600, 64 spp        (mid-size: 105 k triangles, depth-26 tree)
  c4     synthetic ~10M-triangle height field under area lights, 1024x1024, 16 spp (config 4)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "cuda-path-tracer_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def make_c4(n_side, width, height, spp):
    """Synthetic BVH-traversal stress scene: a jittered, displaced n_side x n_side-quad height field
    (2*n_side^2 triangles; 2236 -> 10.0 M) inside cbox-like walls under four quad lights, all diffuse 0.5."""
    import nori_hip as NH
    import oracle_lib as O
    g = n_side + 1
    f = O.pcg32_floats(1234, 1, 3 * g * g).reshape(g, g, 3)          # vertices from pcg32.seed(1234, 1)
    step = 2.0 / n_side
    xs, zs = np.meshgrid(np.linspace(-1, 1, g, dtype=np.float32), np.linspace(-1, 1, g, dtype=np.float32))
    jit = (f[:, :, 0:2] - 0.5) * 0.6 * step
    jit[0, :, :] = 0; jit[-1, :, :] = 0; jit[:, 0, :] = 0; jit[:, -1, :] = 0
    V = np.stack([xs + jit[:, :, 0], 0.2 * f[:, :, 2], zs + jit[:, :, 1]], 2).reshape(-1, 3).astype(np.float32)
    i = np.arange(g * g, dtype=np.uint32).reshape(g, g)
    a, b, c, d = i[:-1, :-1].ravel(), i[:-1, 1:].ravel(), i[1:, 1:].ravel(), i[1:, :-1].ravel()
    F = np.concatenate([np.stack([a, c, b], 1), np.stack([a, d, c], 1)]).astype(np.uint32)
    sb = NH.SceneBuilder()
    sb.add_mesh("heightfield", V, F, color=(0.5, 0.5, 0.5))
    # closed box: back, left, right, ceiling, front (the floor is the height field)
    def quad(p):
        return np.array(p, np.float32), np.array([[0, 1, 2], [0, 2, 3]], np.uint32)
    walls = {
        "back": [[-1, 0, -1], [1, 0, -1], [1, 2, -1], [-1, 2, -1]],
        "front": [[1, 0, 1], [-1, 0, 1], [-1, 2, 1], [1, 2, 1]],
        "left": [[-1, 0, 1], [-1, 0, -1], [-1, 2, -1], [-1, 2, 1]],
        "right": [[1, 0, -1], [1, 0, 1], [1, 2, 1], [1, 2, -1]],
        "ceil": [[-1, 2, -1], [1, 2, -1], [1, 2, 1], [-1, 2, 1]],
    }
    for k, p in walls.items():
        v, fi = quad(p)
        sb.add_mesh(k, v, fi, color=(0.5, 0.5, 0.5))
    for li, (cx, cz) in enumerate([(-0.5, -0.5), (0.5, -0.5), (0.5, 0.5), (-0.5, 0.5)]):
        s = 0.15
        v, fi = quad([[cx - s, 1.98, cz - s], [cx - s, 1.98, cz + s], [cx + s, 1.98, cz + s], [cx + s, 1.98, cz - s]])
        sb.add_mesh("light%d" % li, v, fi, color=(0.5, 0.5, 0.5), radiance=(15, 15, 15))
    sb.set_camera((0.0, 1.0, 0.95), (0.0, 0.1, -0.3), (0, 1, 0), 60.0, width, height)
    return sb.finish(spp=spp)


WORKLOADS = {
    "cbox": dict(xml="scenes/cbox/cbox_path_mis.xml", width=512, height=512, spp=64,
                 name="cbox path_mis 512x512 64spp (BASELINE config 1 geometry, 10252 triangles)"),
    "table": dict(xml="scenes/table/table_path_mis_microfacet.xml", width=800, height=600, spp=256,
                  name="table path_mis microfacet+dielectric 800x600 256spp (BASELINE config 2)"),
    "dragon": dict(xml="scenes/proj/dragon-diffuse.xml", width=800, height=600, spp=64,
                   name="dragon-diffuse path_mis 800x600 64spp (scenes/proj, 105k triangles, mesh emitter)"),
    # BASELINE config 3 (proj scene, 1920x1080, 1024 spp): scenes/proj/final.xml cannot be loaded even by the reference (missing
    # meshes); textureAndBump.xml is the proj scene with every A16 feature whose assets exist
    "proj": dict(xml="scenes/proj/textureAndBump.xml", width=1920, height=1080, spp=1024,
                 name="proj textureAndBump path_mis 1920x1080 1024spp (BASELINE config 3: sphere, image_color, bump_map)"),
    "subsurface": dict(xml="scenes/proj/cbox_subsurface.xml", width=800, height=600, spp=50,
                       name="proj cbox_subsurface path_mis 800x600 50spp as shipped (two subsurface shapes, probe rays)"),
    "c4": dict(xml=None, width=1024, height=1024, spp=16, n_side=2236,
               name="synthetic 10M-triangle height field under 4 area lights 1024x1024 16spp (BASELINE config 4)"),
}


def film_tensor(ctx, torch):
    """torch view of the context's film (no copy), for the RCCL all-reduce"""
    ptr, nbytes = ctx.film_device_ptr()

    class _Film:
        __cuda_array_interface__ = {"shape": (nbytes // 4,), "typestr": "<f4", "data": (ptr, False), "version": 2}
    return torch.as_tensor(_Film(), device="cuda")


def cpu_baseline(w, args):
    """The oracle (CPU restatement, pinned bit-for-bit against the reference's sources) timed on this host: all threads the
    box gives us on a bounded sample of the workload, one thread on a smaller one, BVH build apart."""
    import oracle_lib as O
    threads = min(16, os.cpu_count() or 1)
    if w["xml"] is None:
        return None
    spp = args.cpu_spp
    t0 = time.time()
    o = O.OracleScene(os.path.join(ROOT, w["xml"]), w["width"], w["height"], spp)
    load_s = time.time() - t0                                   # OBJ parse + SAH BVH build (one thread, the reference's order)
    sec, st = o.render(0, spp, mode=0, threads=threads)
    n = w["width"] * w["height"] * spp
    spp1 = max(1, spp // 16)
    sec1, _ = o.render(0, spp1, mode=0, threads=1)
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": round(n / sec / 1e6, 4), "unit": "Msamples/s", "cores": threads, "kind": "port",
            "sample": "%dx%d, %d spp of the same scene (%.1f s), reference block-stream mode; the unmodified reference measured 1.62 Msamples/s on 8 threads in the survey container" % (
                w["width"], w["height"], spp, sec),
            "mrays_per_s": round((st["closest"] + st["shadow"]) / sec / 1e6, 3),
            "single_thread": {"value": round(w["width"] * w["height"] * spp1 / sec1 / 1e6, 4), "sample": "%d spp (%.1f s)" % (spp1, sec1)},
            "cpu_model": model, "host_cpus": os.cpu_count(), "scene_load_and_bvh_build_s": round(load_s, 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="table", choices=list(WORKLOADS))
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--cpu-spp", type=int, default=192)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--c4-side", type=int, default=0)
    ap.add_argument("--partition", default="spp", choices=["spp", "tiles"],
                    help="spp: every rank renders its own spp slice of the whole image (weak scaling, default); tiles: the ranks share one image by interleaved 32-row bands (strong scaling)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    import torch
    import nori_hip as NH
    from nori_hip.partition import spp_slice, tile_bands, allreduce_film
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    os.environ["NORI_HIP_DEVICE"] = str(local)

    w = dict(WORKLOADS[args.workload])
    if args.spp:
        w["spp"] = args.spp
    t0 = time.time()
    if w["xml"] is None:
        hs = make_c4(args.c4_side or w["n_side"], w["width"], w["height"], w["spp"])
    else:
        hs = NH.HostScene(os.path.join(ROOT, w["xml"]), w["width"], w["height"], w["spp"])
    load_s = time.time() - t0
    info = hs.info()
    T = hs.tables()
    info_ext = len(T["textures"]) > 0 or bool((T["shapes"]["bsdf"] == 4).any())      # images or subsurface shapes: the EXT kernels run
    ctx = hs.preprocess(device=local)
    bi = ctx.build_info()
    film = film_tensor(ctx, torch) if world > 1 else None
    spp = w["spp"]
    tiles = args.partition == "tiles"
    samples_per_step = w["width"] * w["height"] * spp                 # per rank (spp slices) or per job (tiles)

    def step(k):
        ctx.clear_film()
        if tiles:
            for y0, y1 in tile_bands(rank, world, w["height"]):      # this rank's bands, all samples
                ctx.render(0, spp, tile=(0, y0, w["width"], y1))
        else:
            ctx.render(*spp_slice(rank, world, spp))      # this rank's spp slice
        if film is not None:
            ctx.synchronize()                             # film complete before the collective reads it
            allreduce_film(film, dist)

    def sync():
        ctx.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for k in range(args.warmup):
        step(k)
    sync()
    ctx.reset_stats()
    kernel_ms = []
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
        if world == 1 and k == args.steps - 1:
            pass
    sync()
    dt = time.perf_counter() - t0
    st = ctx.stats()
    # dominant kernel: the extend stage (k_trace_pool<closest>) of the wavefront pipeline, timed by the library with
    # HIP events around every one of its launches on the context's stream; with NORI_HIP_PIPELINE=mega the single
    # persistent kernel (k_render_w4), timed the same way around its last launch
    wavefront = st["extend_launches"] > 0
    if wavefront:
        launches = st["extend_launches"]
        kernel_ms = st["extend_ms"] / launches
        k_rays, k_boxes, k_prims = st["extend_rays"] / launches, st["extend_boxes"] / launches, st["extend_prims"] / launches
    else:
        kernel_ms = st["render_ms"]
    if dist is not None:
        tmax = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    total_samples = samples_per_step * args.steps * (1 if tiles else world)
    rays = (st["closest_rays"] + st["shadow_rays"]) / args.steps            # this rank, per step
    # algorithmic bytes per step: 32 B per node box tested + 40 B per primitive tested
    # + 44 B ray in + 16 B hit out per ray + 2 x 16 B film read-modify-write per camera sample
    if wavefront:
        alg_bytes = 32.0 * k_boxes + 40.0 * k_prims + 60.0 * k_rays              # per extend launch
        kname = "k_trace_pool<TK_BOTH>" if os.environ.get("NORI_HIP_MERGE", "1") != "0" else "k_trace_pool<TK_EXTEND>"
    else:
        alg_bytes = (32.0 * st["boxes_tested"] + 40.0 * st["prims_tested"]) / args.steps + 60.0 * rays + 32.0 * samples_per_step
        kname = "k_render_ext" if info_ext else "k_render_w%s" % os.environ.get("NORI_HIP_MEGA_WAVES", "4")
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
    # HBM bytes per launch of the same kernel from the PMC passes (FETCH_SIZE x2 + WRITE_SIZE, scratch/pmc.sh), which
    # cannot run inside this process: the committed measurement of this workload and pipeline, if there is one
    traffic, issue = None, None
    tpath = os.path.join(ROOT, "profiles", "r01_%s_traffic.json" % args.workload)
    if os.path.exists(tpath) and not args.spp:
        tj = json.load(open(tpath))
        if tj.get("kernel") == kname:
            traffic = int(tj["traffic_bytes_per_launch"])
            issue = tj.get("valu_issue_busy")
    if traffic is not None and traffic < 0.5 * alg_bytes:
        roof_note = ("algorithmic bytes (SURVEY 8d) count every box and record test at full size; on this scene the nodes and records stay in L1/L2 "
                     "(traffic = HBM bytes per launch from the committed PMC passes), so frac can pass 1: what bounds the kernel is vector-ALU issue "
                     "under divergence (valu_issue_busy, same passes), see DESIGN.md 5")
    else:
        roof_note = "algorithmic bytes per SURVEY 8d; traffic = HBM bytes per launch from the committed PMC passes (null: not measured for this workload and pipeline)"
    out = {
        "metric": "Msamples/s (path_mis, whole node)", "value": round(total_samples / dt / 1e6, 3), "unit": "Msamples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "strong" if tiles else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic" if w["xml"] is None else "reference scene files (scenes/)",
        "config": {"workload": w["name"], "width": w["width"], "height": w["height"], "spp_per_gpu": spp, "triangles": info["prims"],
                   "bvh_nodes_64B": info["wide_nodes"], "partition": ("interleaved 32-row bands" if tiles else "spp slices") + ", RCCL all-reduce of the fp32 film" if world > 1 else "single GPU",
                   "scene_load_s": round(load_s, 2), "bvh_build_s": round(hs.bvh_build_seconds(), 3),
                   "bvh": ("device LBVH, %.1f ms, %d of %d nodes used, depth %d" % (bi["build_ms"], bi["n_used"], bi["n_nodes"], bi["depth"])) if bi["n_nodes"] or bi["build_ms"] else "host SAH (reference order)"},
        "mrays_per_s": round(rays * world * args.steps / dt / 1e6, 2),
        "rays_per_sample": round(rays / samples_per_step, 3),
        "roofline": {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "valu_issue_busy": issue,
                     "note": roof_note,
                     "kernel_ms": round(kernel_ms, 4), "launches_per_step": (round(st["extend_launches"] / args.steps, 1) if wavefront else 1), "algorithmic_bytes_per_launch": int(alg_bytes),
                     "boxes_tested_per_ray": round(st["boxes_tested"] / args.steps / rays, 2), "prims_tested_per_ray": round(st["prims_tested"] / args.steps / rays, 2)},
    }
    if not args.no_cpu and world == 1:
        cb = cpu_baseline(w, args)
        if cb:
            out["cpu_baseline"] = cb
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
