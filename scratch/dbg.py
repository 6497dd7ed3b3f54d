import os, sys, numpy as np
sys.path.insert(0, "cuda-path-tracer_amd"); sys.path.insert(0, "tests")
import nori_hip as NH
g = np.load("tests/golden/cbox_path_mats.npz")
hs = NH.HostScene("scenes/cbox/cbox_path_mats.xml", 64, 64, 1); ctx = hs.preprocess()
h = ctx.trace(g["rays"]); hu = np.frombuffer(h.tobytes(), np.uint32).reshape(-1, 4)
d = (hu != g["hits"])
print("differing per column", d.sum(0), "rows", d.any(1).sum())
i = np.where(d.any(1))[0][:5]
print(hu[i], g["hits"][i], h[i])
