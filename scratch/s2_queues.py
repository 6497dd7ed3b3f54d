#!/usr/bin/env python3
"""Why is cbox with two pipelines slower after another context has rendered with two pipelines in the same process?
usage: s2_queues.py <mode>   mode: alone | after_c4 | after_c4_closed | after_cbox | after_cbox_closed"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-path-tracer_amd")); sys.path.insert(0, ROOT)
import nori_hip as NH
from nori_hip import synthetic
mode = sys.argv[1]
def cbox():
    os.environ["NORI_HIP_BVH"] = "host"
    hs = NH.HostScene(os.path.join(ROOT, "scenes/cbox/cbox_path_mis.xml"), 512, 512, 64)
    return hs, hs.preprocess(device=0)
def measure(ctx, spp, steps=5):
    ctx.clear_film(); ctx.render(0, spp); ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.clear_film(); ctx.render(0, spp)
    ctx.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
first = None
if mode.startswith("after_c4"):
    os.environ["NORI_HIP_BVH"] = "device"
    first = synthetic.make_c4(600, 512, 512, 4); c = first.preprocess(device=0)
    print("first (small c4) %.2f ms" % measure(c, 4, 2), "pipelines", c.stats()["pipelines"])
elif mode.startswith("after_cbox"):
    fp = os.environ.get("FIRST_PIPELINES")
    if fp: os.environ["NORI_HIP_PIPELINES"] = fp
    first, c = cbox()
    if fp: del os.environ["NORI_HIP_PIPELINES"]
    print("first (cbox) %.2f ms" % measure(c, 64), "pipelines", c.stats()["pipelines"])
if first is not None and mode.endswith("closed"):
    first.close()
hs, ctx = cbox()
print(mode, "cbox %.2f ms" % measure(ctx, 64), "pipelines", ctx.stats()["pipelines"], flush=True)
