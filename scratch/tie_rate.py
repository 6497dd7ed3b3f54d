"""how often the device-built tree names a different primitive than the reference fixtures (always a tie in t)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-path-tracer_amd"))
import nori_hip as NH
os.environ["NORI_HIP_BVH"] = "device"
for name, xml in (("cbox", "cbox/cbox_path_mis.xml"), ("table", "table/table_path_mis.xml"), ("dragon", "proj/dragon-diffuse.xml"), ("cbox_subsurface", "proj/cbox_subsurface.xml")):
    g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    hs = NH.HostScene(os.path.join(ROOT, "scenes", xml), int(g["width"]), int(g["height"]), 1)
    ctx = hs.preprocess()
    tot = diff = 0
    for rays, ref in ((g["rays"], g["hits"]), (g["bounce_rays"], g["bounce_hits"])):
        h = ctx.trace(rays)
        d = h["prim"] != ref[:, 0]
        assert np.array_equal(h["t"].view(np.uint32)[d], ref[d, 1])
        tot += len(rays); diff += int(d.sum())
    print("%-16s %d rays, %d name another primitive (same t)  build %s" % (name, tot, diff, ctx.build_info()))
