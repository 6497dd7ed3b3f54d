import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-path-tracer_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import nori_hip as NH
g = np.load(os.path.join(ROOT, "tests/golden/test_scenes.npz"))
def bad(ctx, key, rays=None):
    rays = g[key + "_rays"] if rays is None else rays
    h = ctx.trace(rays)
    hu = np.stack([h["prim"], h["t"].view(np.uint32), h["u"].view(np.uint32), h["v"].view(np.uint32)], 1)
    b = np.nonzero((hu != g[key + "_hits"]).any(1))[0]
    runs = []
    for x in b:
        if runs and runs[-1][1] == x - 1: runs[-1][1] = x
        else: runs.append([x, x])
    return len(b), runs[:8]
ht = NH.HostScene(os.path.join(ROOT, "scenes/tests/test-furnace.xml"))
c2 = ht.preprocess(2); c3 = ht.preprocess(3)
print("ctx2 rays2", bad(c2, "test_furnace_2")); print("ctx3 rays3", bad(c3, "test_furnace_3")); print("ctx3 rays3 again", bad(c3, "test_furnace_3"))
print("ctx2 rays3", bad(c2, "test_furnace_3")); print("ctx3 rays2", bad(c3, "test_furnace_2"))
r3 = g["test_furnace_3_rays"]
print("ctx3 rays3 reversed order", bad(c3, "test_furnace_3")[0], "->", end=" ")
h = c3.trace(r3[::-1].copy())[::-1]
hu = np.stack([h["prim"], h["t"].view(np.uint32), h["u"].view(np.uint32), h["v"].view(np.uint32)], 1)
b = np.nonzero((hu != g["test_furnace_3_hits"]).any(1))[0]; print(len(b), b[:3], b[-3:] if len(b) else "")
print(ht.info())
