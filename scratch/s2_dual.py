#!/usr/bin/env python3
"""Two contexts on one GPU, each rendering half of the samples on its own stream, against one context rendering all of them: does the
trace stage of one pipeline (vector-ALU bound on scenes that live in the caches) overlap with the shade stage of the other (bound by
memory latency)?  usage: s2_dual.py <workload> [steps]"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-path-tracer_amd")); sys.path.insert(0, ROOT)
import nori_hip as NH
from nori_hip import synthetic
import bench
wl = sys.argv[1]; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
N = int(sys.argv[3]) if len(sys.argv) > 3 else 2
w = bench.WORKLOADS[wl]
bvh = "device" if w["xml"] is None else "host"
os.environ["NORI_HIP_BVH"] = bvh
def scene():
    if w["xml"] is None:
        return synthetic.make_c4(w["n_side"], w["width"], w["height"], w["spp"])
    return NH.HostScene(os.path.join(ROOT, w["xml"]), w["width"], w["height"], w["spp"])
spp = w["spp"]
hs = [scene() for _ in range(N)]
ctx = [h.preprocess(device=0) for h in hs]
def timed(fn):
    fn(); 
    t0 = time.perf_counter()
    for _ in range(steps): fn()
    return (time.perf_counter() - t0) / steps * 1e3
def single():
    ctx[0].clear_film(); ctx[0].render(0, spp); ctx[0].synchronize()
def dual():
    def run(i, b, e):
        ctx[i].clear_film(); ctx[i].render(b, e); ctx[i].synchronize()
    th = [threading.Thread(target=run, args=(i, spp * i // N, spp * (i + 1) // N)) for i in range(N)]
    for t in th: t.start()
    for t in th: t.join()
def halves_serial():
    for i in range(N):
        ctx[i].clear_film(); ctx[i].render(spp * i // N, spp * (i + 1) // N); ctx[i].synchronize()
a = timed(single); b = timed(dual); c = timed(halves_serial)
print("%s: one context %.2f ms, %d contexts side by side %.2f ms, their shares one after the other %.2f ms" % (wl, a, N, b, c), flush=True)
