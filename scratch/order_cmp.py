import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-path-tracer_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, nori_hip as NH
for name, xml in [("cbox", "cbox/cbox_path_mis.xml"), ("table", "table/table_path_mis_microfacet.xml"), ("dragon", "proj/dragon-diffuse.xml")]:
    g = np.load(os.path.join(ROOT, "tests/golden", ("table_microfacet" if name == "table" else name) + ".npz"))
    res = {}
    for mode in ("ref", "near"):
        if mode == "near": os.environ["NORI_HIP_ORDER"] = "near"
        hs = NH.HostScene(os.path.join(ROOT, "scenes", xml), int(g["width"]), int(g["height"]), 1); ctx = hs.preprocess()
        os.environ.pop("NORI_HIP_ORDER", None)
        rng = np.random.default_rng(3)
        cam = ctx.camera_rays((rng.random((400000, 2)) * [int(g["width"]), int(g["height"])]).astype(np.float32))
        rays = np.concatenate([np.asarray(g["rays"]).view(np.float32).reshape(-1, 8), np.asarray(g["bounce_rays"]).view(np.float32).reshape(-1, 8), np.asarray(cam).view(np.float32).reshape(-1, 8)])
        ctx.reset_stats()
        h = ctx.trace(rays)
        res[mode] = (np.frombuffer(h.tobytes(), np.uint32).reshape(-1, 4).copy(), ctx.stats())
        hs.close()
    a, b = res["ref"][0], res["near"][0]
    diff = np.any(a != b, axis=1)
    print(name, "rays", len(a), "differing hits", int(diff.sum()), "prim differs", int((a[:, 0] != b[:, 0]).sum()))
    if diff.any():
        i = np.nonzero(diff)[0][:5]
        print("   ref ", a[i].tolist()); print("   near", b[i].tolist())
