import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-path-tracer_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import nori_hip as NH
g = np.load(os.path.join(ROOT, "tests/golden/test_scenes.npz"))
def runs_of(b):
    runs = []
    for x in b:
        if runs and runs[-1][1] == x - 1: runs[-1][1] = int(x)
        else: runs.append([int(x), int(x)])
    return runs
for rep in range(3):
    for fname in ("test-direct.xml", "test-furnace.xml"):
        ht = NH.HostScene(os.path.join(ROOT, "scenes/tests", fname))
        for i in range(ht.scene_count):
            if not ht.test_meta(i)["path_mis"]:
                continue
            ctx = ht.preprocess(i)
            key = "%s_%d" % (fname.replace(".xml", "").replace("-", "_"), i)
            rays = g[key + "_rays"]
            for k in range(2):
                h = ctx.trace(rays)
                hu = np.stack([h["prim"], h["t"].view(np.uint32), h["u"].view(np.uint32), h["v"].view(np.uint32)], 1)
                b = np.nonzero((hu != g[key + "_hits"]).any(1))[0]
                if len(b): print(rep, key, "call", k, "mismatches", len(b), runs_of(b)[:6], "got", h[b[0]], flush=True)
        ht.close()
    print("rep", rep, "done", flush=True)
