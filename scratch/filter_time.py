import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-path-tracer_amd"))
import numpy as np, nori_hip as NH
hs = NH.HostScene(os.path.join(ROOT, "scenes/table/table_path_mis_microfacet.xml"), 800, 600, 16)
ctx = hs.preprocess()
ctx.render(0, 16); ctx.synchronize()
for k in range(3):
    t = time.perf_counter(); out = ctx.filter(r=3, f=10, sigma=0.5, h=0.002); dt = time.perf_counter() - t
    print("filter r=3 f=10 800x600: %.2f ms (host wall, incl. readback)  stats render_ms=%.3f" % (dt * 1e3, ctx.stats()["render_ms"]))
