#!/usr/bin/env python3
"""bench.py — Msamples/s of the path_mis hot path on MI355X.

A "step" is one full render pass of the workload: every pixel, every sample, film accumulation included, with
the scene resident in HBM.  At N > 1 GPUs (one process per GPU, launched by torch.distributed.run) every rank
renders its own spp slice of the same image (weak scaling: per-GPU work is fixed) and the fp32 film is summed
over ranks with one RCCL all-reduce per step, inside the timed region.

Workloads (BASELINE.json configs):
  c4     synthetic 10.0 M-triangle height field under area lights, 1024x1024, 16 spp (configs[3]; default: BASELINE's metric is
         quoted on "cbox & 10M-tri", and this is the one whose working set (1.1 GB of nodes and records) does not fit the caches,
         so the HBM roofline of the traversal kernel is physical on it)
  cbox   scenes/cbox/cbox_path_mis.xml at 512x512, 64 spp      (configs[0], the reference's CPU-runnable case)
  table  scenes/table/table_path_mis_microfacet.xml 800x600, 256 spp   (configs[1])
  dragon scenes/proj/dragon-diffuse.xml 800x600, 64 spp        (mid-size: 105 k triangles, depth-26 tree)
  proj / subsurface   scenes/proj (configs[2])
  c5     cbox 3840x2160 split by spp, RCCL reduce, filter pass (configs[4]; --spp sets the per-GPU sample count)
--bvh host|device: the reference's SAH tree in the reference's order (bit-exact hit ids on ties; built on the host) or the tree
built on the GPU (default for c4, whose host build takes 14 s).
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # before anything starts the HIP runtime: see nori_hip/__init__.py
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "cuda-path-tracer_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


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
    # BASELINE config 5: 16384 spp over 8 GPUs = 2048 spp per GPU; the default here is a scaled per-GPU count (same code path) so
    # that a default run finishes in seconds, --spp 2048 is the configuration as stated
    "c5": dict(xml="scenes/cbox/cbox_path_mis.xml", width=3840, height=2160, spp=32, filter=True,
               name="cbox path_mis 3840x2160 split by spp, RCCL film reduce, NL-means filter pass (BASELINE config 5; spp per GPU scaled from 2048)"),
}
FILTER_PARAMS = dict(r=3, f=10, sigma=0.5, h=0.002)      # the CUDA fork's scene defaults (cuda/nori/scene.cpp:28-32)


def film_tensor(ctx, torch):
    """torch view of the context's film (no copy), for the RCCL all-reduce"""
    ptr, nbytes = ctx.film_device_ptr()

    class _Film:
        __cuda_array_interface__ = {"shape": (nbytes // 4,), "typestr": "<f4", "data": (ptr, False), "version": 2}
    return torch.as_tensor(_Film(), device="cuda")


def committed_traffic(args, kname, bvh, launches_per_step=None):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes of this workload, tree and kernel (they cannot run
    inside this process): profiles/r03_<workload>_traffic.json, else round 2's; (None, None) when none matches (a file whose launch
    count per step differs was taken with another schedule - one pipeline against two, say - and its launches are not these)"""
    if args.spp or args.c4_side:
        return None, None
    for rnd in ("r03", "r02"):
        tpath = os.path.join(ROOT, "profiles", "%s_%s_traffic.json" % (rnd, args.workload))
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            if launches_per_step is not None and tj.get("launches") and abs(tj["launches"] - launches_per_step) > 0.05 * launches_per_step:
                continue
            if tj.get("kernel") == kname and tj.get("bvh", bvh) == bvh:
                return int(tj["traffic_bytes_per_launch"]), tj.get("valu_issue_busy")
    return None, None


def roofline_dict(kname, achieved, kernel_ms, traffic, issue, note, extra):
    """`achieved` / `frac`: ALGORITHMIC bytes per launch over the launch's duration, against the HBM peak (SURVEY 8d's definition: work
    retired, caches included).  `traffic_gbs` / `traffic_frac`: the bytes the memory side actually moved per launch (PMC counters of the
    committed profile) over the same duration - how busy the HBM pins are.  Where the two differ, the difference is reuse in L2."""
    d = {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": round(achieved / HBM_PEAK_GBS, 4), "achieved_algorithmic": round(achieved, 1), "traffic": traffic,
         "traffic_gbs": (round(traffic / (kernel_ms * 1e-3) / 1e9, 1) if traffic is not None else None),
         "traffic_frac": (round(traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic is not None else None),
         "valu_issue_busy": issue, "note": note, "kernel_ms": round(kernel_ms, 4)}
    d.update(extra)
    return d


def load_workload(args, NH, synthetic):
    """(workload dict, bvh mode, host scene, load seconds) for args.workload; NORI_HIP_BVH is left set to the mode"""
    w = dict(WORKLOADS[args.workload])
    if args.spp:
        w["spp"] = args.spp
    bvh = args.bvh or os.environ.get("NORI_HIP_BVH") or ("device" if w["xml"] is None else "host")
    os.environ["NORI_HIP_BVH"] = bvh
    t0 = time.time()
    if w["xml"] is None:
        hs = synthetic.make_c4(args.c4_side or w["n_side"], w["width"], w["height"], w["spp"])
    else:
        hs = NH.HostScene(os.path.join(ROOT, w["xml"]), w["width"], w["height"], w["spp"])
    return w, bvh, hs, time.time() - t0


def group_mode(args):
    """--driver group: the product's own multi-GPU path.  One process, nori_hip_group_render over min(--gpus, devices present)
    devices: every member renders its spp slice (weak scaling: --gpus x spp samples per pixel in all) or its 32-row bands
    (--partition tiles, strong), the films are summed in place by one RCCL all-reduce inside the call, and for the c5 workload
    the filter pass follows on member 0.  Same JSON line as the default mode; phases_ms_per_step.reduce is the device time of the
    all-reduce (nori_hip_group_reduce_ms)."""
    import nori_hip as NH
    from nori_hip import synthetic
    w, bvh, hs, load_s = load_workload(args, NH, synthetic)
    n_dev = max(1, min(args.gpus, NH.device_count()))
    g = NH.Group(list(range(n_dev)))
    g.upload(hs.flatten(), build_on_device=(bvh == "device"))
    c0 = g.context(0)
    tiles = args.partition == "tiles"
    spp_total = w["spp"] if tiles else w["spp"] * n_dev
    part = NH.PARTITION_TILES if tiles else NH.PARTITION_SPP
    do_filter = bool(w.get("filter"))
    phase = {"render": 0.0, "reduce": 0.0, "filter": 0.0}

    def step(timed):
        g.clear_film()
        t_a = time.perf_counter()
        g.render(0, spp_total, part)                      # returns when every member holds the summed film
        t_b = time.perf_counter()
        if do_filter:
            c0.filter(**FILTER_PARAMS)
        t_c = time.perf_counter()
        if timed:
            red = g.reduce_ms() * 1e-3
            phase["render"] += (t_b - t_a) - red; phase["reduce"] += red; phase["filter"] += t_c - t_b

    for _ in range(args.warmup):
        step(False)
    for i in range(n_dev):
        g.context(i).synchronize(); g.context(i).reset_stats()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    for i in range(n_dev):
        g.context(i).synchronize()
    dt = time.perf_counter() - t0
    sts = [g.context(i).stats() for i in range(n_dev)]
    st = sts[0]
    rays_job = float(sum(x["closest_rays"] + x["shadow_rays"] for x in sts))
    samples = w["width"] * w["height"] * spp_total * args.steps
    launches = max(1, st["extend_launches"])
    kernel_ms = st["extend_ms"] / launches
    if st["extend_width"] == 4 and st["extend_node_bytes"] == 64:              # quantised four-box nodes: 64 B per visit (see main())
        alg_bytes = (64.0 * st["extend_visits"] + 40.0 * st["extend_prims"] + 60.0 * st["extend_rays"]) / launches
    else:
        alg_bytes = (32.0 * st["extend_boxes"] + 40.0 * st["extend_prims"] + 60.0 * st["extend_rays"]) / launches
    kname = c0.trace_kernel_name()
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9 if st["extend_launches"] else 0.0
    traffic, issue = committed_traffic(args, kname, bvh) if n_dev == 1 else (None, None)
    info = hs.info()
    out = {
        "metric": "Msamples/s (path_mis, whole node)", "value": round(samples / dt / 1e6, 3), "unit": "Msamples/s",
        "n_gpus": n_dev, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "strong" if tiles else "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic" if w["xml"] is None else "reference scene files (scenes/)",
        "config": {"workload": w["name"], "width": w["width"], "height": w["height"], "spp_per_gpu": w["spp"], "triangles": info["prims"],
                   "driver": "group: one process, nori_hip_group_render over %d device(s) (%d asked for)" % (n_dev, args.gpus),
                   "partition": ("32-row bands, one render call per member" if tiles else "spp slices") + ", in-place RCCL all-reduce of the fp32 film",
                   "scene_load_s": round(load_s, 2), "bvh": "device-built" if bvh == "device" else "host SAH (reference order)"},
        "mrays_per_s": round(rays_job / dt / 1e6, 2),
        "roofline": roofline_dict(kname, achieved, kernel_ms, traffic, issue,
                                  "member 0's trace stage; algorithmic bytes per SURVEY 8d, counted by the kernel itself", {"algorithmic_bytes_per_launch": int(alg_bytes)}),
        "phases_ms_per_step": {k: round(v / args.steps * 1e3, 3) for k, v in phase.items()},
    }
    emit(out)
    g.close(); hs.close()


def as_rank_mode(args):
    """--as-rank N: what strong scaling over N GPUs would look like, measured on one.  The whole workload is timed, then the share
    of every rank r of an N-way split - its interleaved 32-row bands in one nori_hip_render_tiles call, and (if spp >= N) its spp
    slice of the whole image - is rendered alone on this GPU.  On a node the ranks run side by side, so a step lasts as long as the
    slowest share plus the film reduce (1.5 ms for 133 MB): predicted efficiency = whole / (N x slowest share)."""
    import nori_hip as NH
    from nori_hip import synthetic
    from nori_hip.partition import split_spp, tile_bands
    w, bvh, hs, load_s = load_workload(args, NH, synthetic)
    ctx = hs.preprocess(device=0)
    N, W, H, spp = args.as_rank, w["width"], w["height"], w["spp"]

    def timed(fn):
        for _ in range(max(1, args.warmup)):
            ctx.clear_film(); fn(); ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ctx.clear_film(); fn()
        ctx.synchronize()
        return (time.perf_counter() - t0) / args.steps * 1e3

    whole = timed(lambda: ctx.render(0, spp))
    out = {"mode": "as-rank", "workload": w["name"], "n_ranks": N, "steps": args.steps, "ms_whole": round(whole, 3), "ms_ideal_share": round(whole / N, 3), "partitions": {}}
    shares = []
    for r in range(N):
        bands = [(0, y0, W, y1) for y0, y1 in tile_bands(r, N, H)]
        shares.append(timed(lambda: ctx.render_tiles(0, spp, bands)) if bands else 0.0)
    out["partitions"]["tiles"] = {"ms_per_rank": [round(x, 3) for x in shares], "ms_slowest": round(max(shares), 3),
                                  "predicted_efficiency": round(whole / (N * max(shares)), 4)}
    if spp >= N:
        shares = [timed(lambda: ctx.render(b, e)) for b, e in split_spp(spp, N)]
        out["partitions"]["spp"] = {"ms_per_rank": [round(x, 3) for x in shares], "ms_slowest": round(max(shares), 3),
                                    "predicted_efficiency": round(whole / (N * max(shares)), 4)}
    emit(out)
    hs.close()


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(w, args):
    """The oracle (CPU restatement, pinned bit-for-bit against the reference's sources) timed on this host: all threads the
    box gives us on a bounded sample of the workload, one thread on a smaller one, BVH build apart."""
    import oracle_lib as O
    threads = min(16, os.cpu_count() or 1)
    if w["xml"] is None:
        # C4: the same mesh (all 10 M triangles unless --cpu-c4-side says otherwise), written as OBJ + XML and loaded by the oracle
        # as the reference would load it (text parse + a one-thread SAH build: about a minute, reported apart), rendered at a
        # reduced image size: the bounded sample is fewer pixels and samples per pixel of the same scene through the same camera (about 20 s
        # on the GPU box's 16 host threads).
        import tempfile
        from nori_hip import synthetic
        side, cw, ch, spp = args.cpu_c4_side or w["n_side"], 256, 256, args.cpu_spp or 8
        with tempfile.TemporaryDirectory() as d:
            t0 = time.time()
            xml = synthetic.write_c4(d, side, cw, ch, spp)
            write_s = time.time() - t0
            t0 = time.time()
            o = O.OracleScene(xml, cw, ch, spp)
            load_s = time.time() - t0
        sec, st = o.render(0, spp, mode=0, threads=threads)
        n = cw * ch * spp
        return {"value": round(n / sec / 1e6, 5), "unit": "Msamples/s", "cores": threads, "kind": "port",
                "sample": "the same scene (%d of %d triangles), %dx%d instead of %dx%d pixels, %d spp (%.1f s), reference block-stream mode" % (
                    o.info()["prims"], 2 * w["n_side"] ** 2 + 18, cw, ch, w["width"], w["height"], spp, sec),
                "mrays_per_s": round((st["closest"] + st["shadow"]) / sec / 1e6, 4),
                "boxes_tested_per_ray": round(st["nodes"] / max(1, st["closest"] + st["shadow"]), 1),
                "cpu_model": cpu_model(), "host_cpus": os.cpu_count(), "obj_write_s": round(write_s, 2), "scene_load_and_bvh_build_s": round(load_s, 2)}
    spp = args.cpu_spp or (192 if w["width"] * w["height"] < 10 ** 6 else 8)
    cw, ch = w["width"], w["height"]
    t0 = time.time()
    o = O.OracleScene(os.path.join(ROOT, w["xml"]), cw, ch, spp)
    load_s = time.time() - t0                                   # OBJ parse + SAH BVH build (one thread, the reference's order)
    sec, st = o.render(0, spp, mode=0, threads=threads)
    n = cw * ch * spp
    spp1 = max(1, spp // 16)
    sec1, _ = o.render(0, spp1, mode=0, threads=1)
    return {"value": round(n / sec / 1e6, 4), "unit": "Msamples/s", "cores": threads, "kind": "port",
            "sample": "%dx%d, %d spp of the same scene (%.1f s), reference block-stream mode; the unmodified reference measured 1.62 Msamples/s on 8 threads in the survey container" % (
                cw, ch, spp, sec),
            "mrays_per_s": round((st["closest"] + st["shadow"]) / sec / 1e6, 3),
            "single_thread": {"value": round(cw * ch * spp1 / sec1 / 1e6, 4), "sample": "%d spp (%.1f s)" % (spp1, sec1)},
            "cpu_model": cpu_model(), "host_cpus": os.cpu_count(), "scene_load_and_bvh_build_s": round(load_s, 3)}


_JSON_FD = None


def emit(out):
    """the ONE JSON line of this file's contract, written to the process's original stdout"""
    line = (json.dumps(out) + "\n").encode()
    if _JSON_FD is None:
        sys.stdout.write(line.decode()); sys.stdout.flush()
    else:
        os.write(_JSON_FD, line)


def main():
    # Native libraries print to stdout as they please (RCCL's version banner at communicator creation): from here on descriptor 1 is
    # the process's stderr, and the JSON line goes to a private duplicate of the original stdout.
    global _JSON_FD
    sys.stdout.flush()
    _JSON_FD = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c4", choices=list(WORKLOADS))
    ap.add_argument("--spp", type=int, default=0, help="samples per pixel per GPU (default: the workload's)")
    ap.add_argument("--bvh", default="", choices=["", "host", "device"], help="default: device for c4, host (the reference's tree and order) otherwise")
    ap.add_argument("--cpu-spp", type=int, default=0)
    ap.add_argument("--cpu-c4-side", type=int, default=0, help="c4 cpu_baseline: quads per side of the height field (default: the workload's own, 2236)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the cbox measurement that the c4 run reports beside its own")
    ap.add_argument("--c4-side", type=int, default=0)
    ap.add_argument("--partition", default="spp", choices=["spp", "tiles"],
                    help="spp: every rank renders its own spp slice of the whole image (weak scaling, default); tiles: the ranks share one image by interleaved 32-row bands (strong scaling)")
    ap.add_argument("--driver", default="ranks", choices=["ranks", "group"],
                    help="ranks: one process per GPU, torch.distributed all-reduce (the launch contract of this file, default); group: ONE process drives --gpus devices "
                         "through nori_hip_group_* (what the path_mis plugin's `devices` property and the CLI use; fewer devices present: as many as there are)")
    ap.add_argument("--as-rank", type=int, default=0, metavar="N",
                    help="strong-scaling prediction on one GPU: render every rank's share of an N-way split of the workload (tiles, and spp slices if spp >= N), "
                         "one after the other, and print the slowest share against whole/N")
    args = ap.parse_args()
    if args.as_rank:
        return as_rank_mode(args)
    if args.driver == "group":
        return group_mode(args)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    import torch
    import nori_hip as NH
    from nori_hip import synthetic
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
    bvh = args.bvh or os.environ.get("NORI_HIP_BVH") or ("device" if w["xml"] is None else "host")
    os.environ["NORI_HIP_BVH"] = bvh
    t0 = time.time()
    if w["xml"] is None:
        hs = synthetic.make_c4(args.c4_side or w["n_side"], w["width"], w["height"], w["spp"])
    else:
        hs = NH.HostScene(os.path.join(ROOT, w["xml"]), w["width"], w["height"], w["spp"])
    load_s = time.time() - t0
    info = hs.info()
    T = hs.tables()
    info_ss = bool((T["shapes"]["bsdf"] == 4).any())
    info_ext = len(T["textures"]) > 0 or info_ss                                      # images or subsurface shapes: the EXT kernels run
    ctx = hs.preprocess(device=local)
    bi = ctx.build_info()
    film = film_tensor(ctx, torch) if world > 1 else None
    spp = w["spp"]
    tiles = args.partition == "tiles"
    do_filter = bool(w.get("filter"))
    samples_per_step = w["width"] * w["height"] * spp                 # per rank (spp slices) or per job (tiles)
    phase = {"render": 0.0, "reduce": 0.0, "filter": 0.0}

    def step(k, timed=False, ctx=ctx, film=film, w=w, spp=spp, do_filter=do_filter):
        ctx.clear_film()
        t_a = time.perf_counter()
        if tiles:
            ctx.render_tiles(0, spp, [(0, y0, w["width"], y1) for y0, y1 in tile_bands(rank, world, w["height"])])   # this rank's bands, ONE call
        else:
            ctx.render(*spp_slice(rank, world, spp))      # this rank's spp slice
        if film is not None or do_filter:
            ctx.synchronize()                             # film complete before the collective / the filter reads it
        t_b = time.perf_counter()
        if film is not None:
            allreduce_film(film, dist)
            torch.cuda.synchronize()
        t_c = time.perf_counter()
        if do_filter and rank == 0:
            # configs[4]: the filter pass runs once, on the reduced film, on rank 0 (11 ms at 4K against seconds of rendering;
            # the other ranks wait at the next barrier) - cuda/filter.cc.cu:125-176 after src/block.cpp:131-140's merge
            ctx.filter(**FILTER_PARAMS)
        t_d = time.perf_counter()
        if timed:
            phase["render"] += t_b - t_a; phase["reduce"] += t_c - t_b; phase["filter"] += t_d - t_c

    def sync(ctx=ctx):
        ctx.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for k in range(args.warmup):
        step(k)
    sync()
    ctx.reset_stats()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k, True)
    sync()
    dt = time.perf_counter() - t0
    st = ctx.stats()
    kname_stage = ctx.trace_kernel_name()
    bvh_build_s = hs.bvh_build_seconds()
    # (the context goes before the next one comes: every context that renders as two pipelines holds two streams, and a process whose
    # streams outnumber its hardware queues has some of them share one - INTEGRATION.md, "two pipelines")
    hs.close()

    # second line of evidence in the same run: BASELINE's metric is quoted on "cbox and the 10 M-triangle scene", so the
    # cbox workload (host tree in the reference's order) is timed too, with the same partition, barriers and max over ranks;
    # it goes into "secondary" of the one JSON line and changes nothing of the primary measurement above
    secondary = None
    if args.workload == "c4" and not args.no_secondary:
        w2 = dict(WORKLOADS["cbox"])
        os.environ["NORI_HIP_BVH"] = "host"
        hs2 = NH.HostScene(os.path.join(ROOT, w2["xml"]), w2["width"], w2["height"], w2["spp"])
        ctx2 = hs2.preprocess(device=local)
        film2 = film_tensor(ctx2, torch) if world > 1 else None
        kw = dict(ctx=ctx2, film=film2, w=w2, spp=w2["spp"], do_filter=False)
        steps2 = max(5, args.steps)
        step(0, False, **kw)
        sync(ctx2)
        t0 = time.perf_counter()
        for k in range(steps2):
            step(k, False, **kw)
        sync(ctx2)
        dt2 = time.perf_counter() - t0
        if dist is not None:
            t2 = torch.tensor([dt2], device="cuda", dtype=torch.float64)
            dist.all_reduce(t2, op=dist.ReduceOp.MAX)
            dt2 = float(t2.item())
        n2 = w2["width"] * w2["height"] * w2["spp"] * steps2 * (1 if tiles else world)
        secondary = [{"workload": w2["name"], "value": round(n2 / dt2 / 1e6, 3), "unit": "Msamples/s", "ms_per_step": round(dt2 / steps2 * 1e3, 3),
                      "steps": steps2, "warmup": 1, "spp_per_gpu": w2["spp"], "bvh": "host SAH (reference order)"}]
        hs2.close()
        os.environ["NORI_HIP_BVH"] = bvh
    # The same workload once more with every emitter sample's shadow ray traced (NORI_HIP_CULL=0), as the reference does it: the
    # library drops the ray when the sample's term is exactly zero (same film bit for bit).  Reported beside the
    # headline so that both numbers come from one run.
    uncut = None
    if args.workload == "c4" and not args.no_secondary:
        os.environ["NORI_HIP_CULL"] = "0"
        try:
            hs3 = synthetic.make_c4(args.c4_side or w["n_side"], w["width"], w["height"], w["spp"])      # (a scene of its own: preprocess() replaces a scene's context)
            ctx3 = hs3.preprocess(device=local)
        finally:
            del os.environ["NORI_HIP_CULL"]
        film3 = film_tensor(ctx3, torch) if world > 1 else None
        kw = dict(ctx=ctx3, film=film3)
        step(0, False, **kw)
        sync(ctx3)
        ctx3.reset_stats()
        t0 = time.perf_counter()
        for k in range(args.steps):
            step(k, False, **kw)
        sync(ctx3)
        dt3 = time.perf_counter() - t0
        st3 = ctx3.stats()
        if dist is not None:
            t3 = torch.tensor([dt3], device="cuda", dtype=torch.float64)
            dist.all_reduce(t3, op=dist.ReduceOp.MAX)
            dt3 = float(t3.item())
        n3 = samples_per_step * args.steps * (1 if tiles else world)
        uncut = {"value": round(n3 / dt3 / 1e6, 3), "unit": "Msamples/s", "ms_per_step": round(dt3 / args.steps * 1e3, 3),
                 "rays_per_sample": round((st3["closest_rays"] + st3["shadow_rays"]) / args.steps / (samples_per_step if not tiles else max(1, samples_per_step // world)), 3),
                 "note": "NORI_HIP_CULL=0: every emitter sample's shadow ray traced, as in the reference"}
        hs3.close()
    # dominant kernel: the trace stage (k_trace_pool) of the wavefront pipeline, timed by the library with
    # HIP events around every one of its launches on the context's stream; with NORI_HIP_PIPELINE=mega the single
    # persistent kernel (k_render_w4), timed the same way around its last launch
    wavefront = st["extend_launches"] > 0
    if wavefront:
        launches = st["extend_launches"]
        kernel_ms = st["extend_ms"] / launches
        k_rays, k_boxes, k_prims = st["extend_rays"] / launches, st["extend_boxes"] / launches, st["extend_prims"] / launches
    else:
        kernel_ms = st["render_ms"]
    rays_rank = float(st["closest_rays"] + st["shadow_rays"])
    if dist is not None:
        tmax = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        rsum = torch.tensor([rays_rank], device="cuda", dtype=torch.float64)
        dist.all_reduce(rsum, op=dist.ReduceOp.SUM)
        rays_job = float(rsum.item())
    else:
        rays_job = rays_rank
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    total_samples = samples_per_step * args.steps * (1 if tiles else world)
    rays = rays_rank / args.steps                                              # this rank, per step
    # algorithmic bytes: 32 B per box tested + 40 B per primitive record tested + 44 B ray in + 16 B hit out per ray
    survey_bytes = None
    if wavefront:
        if st["extend_width"] == 8 or (st["extend_width"] == 4 and st["extend_node_bytes"] == 64):
            # quantised nodes (eight boxes in 128 B, or four in 64 B): a visit reads one node whatever the number of children in use (a box
            # is 6 bytes there, not the reference's 32: counting 32 B per box tested would credit the kernel with bytes nobody needs)
            alg_bytes = float(st["extend_node_bytes"]) * st["extend_visits"] / launches + 40.0 * k_prims + 60.0 * k_rays
            survey_bytes = 32.0 * k_boxes + 40.0 * k_prims + 60.0 * k_rays
        else:
            alg_bytes = 32.0 * k_boxes + 40.0 * k_prims + 60.0 * k_rays          # per trace launch
        kname = kname_stage
    else:
        alg_bytes = (32.0 * st["boxes_tested"] + 40.0 * st["prims_tested"]) / args.steps + 60.0 * rays + 32.0 * samples_per_step
        kname = ("k_render_ext" if info_ss else "k_render_img") if info_ext else "k_render_w%s" % os.environ.get("NORI_HIP_MEGA_WAVES", "4")
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
    # HBM bytes per launch of the same kernel from the PMC passes (2 x FETCH_SIZE + WRITE_SIZE, scratch/pmc.sh; the doubling
    # holds for scattered reads too: scratch/fetchprobe.hip, profiles/r02_fetchprobe.txt), which cannot run inside this
    # process: the committed measurement of this workload, tree and kernel, if there is one
    traffic, issue = committed_traffic(args, kname, bvh, (st["extend_launches"] / args.steps) if wavefront else None)
    if traffic is not None and traffic < 0.5 * alg_bytes:
        roof_note = ("algorithmic bytes (SURVEY 8d) count every box and record test at full size; on this scene the nodes and records stay in L1/L2 "
                     "(traffic = HBM bytes per launch from the committed PMC passes), so frac is not an HBM fraction here: what bounds the kernel is vector-ALU issue "
                     "under divergence (valu_issue_busy, same passes), see DESIGN.md 5")
    else:
        roof_note = "algorithmic bytes per SURVEY 8d, counted by the kernel itself; traffic = HBM bytes per launch from the committed PMC passes (null: not measured for this workload, tree and kernel)"
        if survey_bytes:
            roof_note += ("; this kernel reads quantised nodes, so a node visit is counted at the node's %d bytes and not at SURVEY 8d's 32 B per box tested "
                          "(that count, comparable with earlier rounds, is achieved_at_32_bytes_per_box)" % int(st["extend_node_bytes"]))
    if st.get("pipelines", 1) == 2:
        roof_note += ("; the render call ran as TWO pipelines side by side (DESIGN.md 5, finding 41), each over half of the samples with a pool of its own: a trace launch covers half "
                      "the slots and shares the GPU with the other pipeline's shade stage, ray sort or trace launch while it runs, so its duration - and with it frac, which is "
                      "per launch - is not what the kernel reaches alone (NORI_HIP_PIPELINES=1 runs one pipeline%s); "
                      "whole_step_frac = the algorithmic bytes of all trace launches of a step over the step's time") % (
                          ": frac 0.57 on this workload, 11 % more time per step, profiles/r03_c4_bench_one_pipeline.json" if args.workload == "c4" else "")
    out = {
        "metric": "Msamples/s (path_mis, whole node)", "value": round(total_samples / dt / 1e6, 3), "unit": "Msamples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "strong" if tiles else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic" if w["xml"] is None else "reference scene files (scenes/)",
        "config": {"workload": w["name"], "width": w["width"], "height": w["height"], "spp_per_gpu": spp, "triangles": info["prims"],
                   "partition": (("interleaved 32-row bands, one render call per rank" if tiles else "spp slices") + ", RCCL all-reduce of the fp32 film") if world > 1 else "single GPU",
                   "scene_load_s": round(load_s, 2), "bvh_build_s": round(bvh_build_s, 3),
                   "bvh": ("device-built, %.1f ms, %d nodes, depth %d" % (bi["build_ms"], bi["n_used"], bi["depth"])) if bvh == "device" else "host SAH (reference order), %d 64-byte nodes" % info["wide_nodes"]},
        "mrays_per_s": round(rays_job / dt / 1e6, 2),
        "rays_per_sample": round(rays / (samples_per_step if not tiles else max(1, samples_per_step // world)), 3),
        # emitter samples whose term was exactly zero get no shadow ray; they are in neither figure above
        "culled_shadow_rays_per_sample": round(st["shadow_rays_culled"] / args.steps / (samples_per_step if not tiles else max(1, samples_per_step // world)), 3),
        "roofline": roofline_dict(kname, achieved, kernel_ms, traffic, issue, roof_note, extra={
                     "launches_per_step": (round(st["extend_launches"] / args.steps, 1) if wavefront else 1), "algorithmic_bytes_per_launch": int(alg_bytes),
                     "boxes_tested_per_ray": round(st["boxes_tested"] / max(1.0, rays_rank), 2), "prims_tested_per_ray": round(st["prims_tested"] / max(1.0, rays_rank), 2),
                     "node_width": int(st["extend_width"]), "node_bytes": int(st["extend_node_bytes"]) if wavefront else None,
                     "achieved_at_32_bytes_per_box": (round(survey_bytes / (kernel_ms * 1e-3) / 1e9, 1) if survey_bytes else None), "node_visits_per_ray": (round(st["extend_visits"] / max(1.0, st["extend_rays"]), 2) if wavefront else None),
                     "pipelines": int(st.get("pipelines", 1)),
                     # the trace stage's algorithmic bytes of a whole step over the step's time (everything else a step does counts as time, not as bytes)
                     "whole_step_frac": (round(alg_bytes * st["extend_launches"] / args.steps / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 4) if wavefront else None),
                     "whole_step_frac_at_32_bytes_per_box": (round(survey_bytes * st["extend_launches"] / args.steps / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 4) if (wavefront and survey_bytes) else None)}),
    }
    if secondary:
        out["secondary"] = secondary
    if uncut:
        out["without_zero_term_cull"] = uncut
    if do_filter or world > 1:
        out["phases_ms_per_step"] = {k: round(v / args.steps * 1e3, 3) for k, v in phase.items()}
    if not args.no_cpu and world == 1:
        cb = cpu_baseline(w, args)
        if cb:
            out["cpu_baseline"] = cb
    emit(out)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
