"""GPU: the synthetic 10M-triangle workload of BASELINE config 4 (nori_hip/synthetic.py) and the deep-tree machinery.

Reduced size (n_side 300: 180 018 triangles), written as the reference's own files (XML + OBJ) and loaded by the host
library and by the oracle: hits and Li against the oracle with the reference-order host tree, and the device-built tree
against it.  Full size (n_side 2236: 10.0 M triangles, device-built tree): size-independent properties."""
import os

import numpy as np
import pytest

import nori_hip as NH
import oracle_lib as O
from nori_hip import synthetic

pytestmark = pytest.mark.gpu


def hits_as_u32(h):
    return np.frombuffer(h.tobytes(), np.uint32).reshape(-1, 4)


@pytest.fixture(scope="module")
def c4_small(tmp_path_factory):
    d = tmp_path_factory.mktemp("c4")
    xml = synthetic.write_c4(str(d), 300, 96, 96, 4)
    return xml


def camera_and_bounce_rays(ctx, W, H, n, seed):
    rng = np.random.default_rng(seed)
    ps = (rng.random((n, 2)) * [W, H]).astype(np.float32)
    rays = ctx.camera_rays(ps)
    hits, full = ctx.trace(rays, full=True)
    ok = hits["prim"] != NH.MISS
    # bounce rays: from the hit point into a random direction of the upper hemisphere of the shading normal
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    nrm = full[:, 5:8]
    flip = (d * nrm).sum(1) < 0
    d[flip] = -d[flip]
    b = np.zeros((n, 8), np.float32)
    b[:, 0:3] = full[:, 0:3]; b[:, 3] = 1e-3; b[:, 4:7] = d; b[:, 7] = np.inf
    return rays, b[ok]


def test_c4_reduced_hits_and_li_against_oracle_with_the_reference_order_tree(c4_small):
    W = H = 96
    hs = NH.HostScene(c4_small, W, H, 4)
    ctx = hs.preprocess()
    o = O.OracleScene(c4_small, W, H, 4)
    assert hs.info()["prims"] == 180018 == o.info()["prims"]
    rays, bounce = camera_and_bounce_rays(ctx, W, H, 20000, 3)
    assert np.array_equal(hits_as_u32(ctx.trace(rays)), o.trace(rays))                 # prim id, t, u, v bit-exact
    assert np.array_equal(hits_as_u32(ctx.trace(bounce)), o.trace(bounce))
    sh = bounce.copy(); sh[:, 7] = 0.7
    assert np.array_equal(ctx.trace(sh, any_hit=True)["prim"], o.trace(sh, any_hit=True)[:, 0])
    n = 4096
    st = np.arange(n, dtype=np.uint64)
    li = ctx.li_samples(st, np.zeros(n, np.uint64), rays=rays[:n])
    oli = o.li_rays(rays[:n], st, np.zeros(n, np.uint64))
    assert np.array_equal(li, oli)         # a diffuse scene, the reference-order tree: every sample's Li bit for bit (nd_sincosf, csrc/nori_device.h)
    # image at matched streams
    ctx.clear_film(); ctx.reset_stats(); ctx.render(0, 4)
    img, film = ctx.read_bitmap(), ctx.read_film()
    _, ost = o.render(0, 4, mode=1, threads=8)
    assert np.array_equal(film[:, :, 3], o.film()[:, :, 3])
    rel = np.abs(img.mean((0, 1)) - o.bitmap().mean((0, 1))) / o.bitmap().mean((0, 1))
    assert rel.max() < 2e-3, rel
    s = ctx.stats()
    assert abs(s["closest_rays"] - ost["closest"]) < 0.01 * ost["closest"] and abs(s["shadow_rays"] + s["shadow_rays_culled"] - ost["shadow"]) < 0.01 * ost["shadow"]
    hs.close()


@pytest.mark.parametrize("width,extra", [("4", {}), ("8", {}),
                                         ("4", {"NORI_HIP_LBVH_ALIGN": "1", "NORI_HIP_LBVH_LEAF": "2", "NORI_HIP_ANYORDER": "far", "NORI_HIP_SPEC": "0", "NORI_HIP_FINISH_WAVES": "4"}),
                                         ("4", {"NORI_HIP_LBVH_ALIGN": "1", "NORI_HIP_LBVH_LEAF": "2", "NORI_HIP_ANYORDER": "far", "NORI_HIP_SPEC": "0", "NORI_HIP_FINISH_WAVES": "4",
                                                "NORI_HIP_QUANT": "0"})],
                         ids=["four-box", "eight-box", "full-size-defaults", "full-size-defaults-float-nodes"])
def test_c4_reduced_device_built_tree_agrees_with_the_host_tree(c4_small, width, extra):
    """width 8: the device-built tree as quantised eight-box nodes (NORI_HIP_WIDE=8; boxes only ever enlarged).
    full-size-defaults: what a tree of a million records or more gets by itself - leaves of two records placed so that none
    straddles a 128-byte line (two of eight record positions stay empty), any-hit rays entering the farthest child first, no
    speculative traversal, the tail kernel built for four waves per SIMD, and the
    trace stage and the tail kernel over 64-byte quantised four-box nodes with six-byte stack entries (k_trace_pool4q_leaf2,
    k_finish<false, 4, 4, true>); -float-nodes: the same tree through the 128-byte nodes (k_trace_pool4_leaf2)."""
    W = H = 96
    hs = NH.HostScene(c4_small, W, H, 4)
    ctx = hs.preprocess()
    env = {"NORI_HIP_BVH": "device", "NORI_HIP_WIDE": width, "NORI_HIP_SORT": "1", **extra}      # (rays ordered and dealt by XCD, as at full size)
    os.environ.update(env)
    try:
        hd = NH.HostScene(c4_small, W, H, 4)
        cd = hd.preprocess()
    finally:
        for k in env:
            del os.environ[k]
    bi = cd.build_info()
    assert bi["n_used"] > 0 and 0 < bi["depth"] <= 62
    rays, bounce = camera_and_bounce_rays(ctx, W, H, 20000, 4)
    for r in (rays, bounce):
        a, b = ctx.trace(r), cd.trace(r)
        same = a["prim"] == b["prim"]
        # another primitive only where two tie in t (shared edges of the height field), never another distance
        assert same.mean() > 0.995, same.mean()
        assert np.array_equal(a["t"][same], b["t"][same])
        assert np.allclose(a["t"][~same], b["t"][~same], rtol=1e-5)
    sh = bounce.copy(); sh[:, 7] = 0.7
    assert np.array_equal(ctx.trace(sh, any_hit=True)["prim"], cd.trace(sh, any_hit=True)["prim"])
    # the two trees render the same image up to the ties
    ctx.clear_film(); ctx.render(0, 4); cd.clear_film(); cd.render(0, 4)
    fa, fb = ctx.read_film(), cd.read_film()
    assert cd.stats()["extend_width"] == int(width)
    if "NORI_HIP_LBVH_ALIGN" in extra:
        assert cd.stats()["extend_node_bytes"] == (128 if extra.get("NORI_HIP_QUANT") == "0" else 64)
        assert cd.trace_kernel_name() == ("k_trace_pool4" if extra.get("NORI_HIP_QUANT") == "0" else "k_trace_pool4q")
    assert np.array_equal(fa[:, :, 3], fb[:, :, 3])
    assert abs(fa[:, :, :3].sum() - fb[:, :, :3].sum()) < 2e-3 * fa[:, :, :3].sum()
    hs.close(); hd.close()


def test_c4_full_size_properties_on_the_device_built_tree():
    """10.0 M triangles: tree depth within the traversal stack, every camera ray hits (closed box), hit distances consistent
    with the hit points, linearity in spp of the film, ray counts proportional to samples."""
    W = H = 256
    os.environ["NORI_HIP_BVH"] = "device"
    try:
        hs = synthetic.make_c4(2236, W, H, 2)
        ctx = hs.preprocess()
    finally:
        del os.environ["NORI_HIP_BVH"]
    assert hs.info()["prims"] == 2 * 2236 * 2236 + 18
    bi = ctx.build_info()
    assert 0 < bi["depth"] <= 62 and bi["n_used"] > 10 ** 6, bi
    rng = np.random.default_rng(11)
    ps = (rng.random((100000, 2)) * [W, H]).astype(np.float32)
    rays = ctx.camera_rays(ps)
    hits, full = ctx.trace(rays, full=True)
    assert (hits["prim"] != NH.MISS).all()                                     # closed scene
    p = rays[:, 0:3] + hits["t"][:, None] * rays[:, 4:7]
    assert np.abs(p - full[:, 0:3]).max() < 2e-3                               # barycentric hit point == o + t d (grazing hits on mm-sized triangles: t carries the error)
    assert (np.abs(full[:, [0, 2]]) <= 1.0 + 1e-5).all() and (full[:, 1] >= -1e-5).all() and (full[:, 1] <= 2.0 + 1e-5).all()
    # any-hit consistent with closest-hit: a ray cut short of its hit is unoccluded, one that passes it is occluded
    sh = rays.copy(); sh[:, 7] = hits["t"] * 0.999; sh[:, 3] = 1e-3
    assert (ctx.trace(sh, any_hit=True)["prim"] == NH.MISS).all()
    sh[:, 7] = hits["t"] * 1.001
    assert (ctx.trace(sh, any_hit=True)["prim"] == 1).all()
    ctx.clear_film(); ctx.reset_stats(); ctx.render(0, 2)
    f2, s2 = ctx.read_film(), ctx.stats()
    ctx.render(2, 4)
    f4, s4 = ctx.read_film(), ctx.stats()
    assert f2[..., 3].sum() == W * H * 2 and f4[..., 3].sum() == W * H * 4 and s4["camera_samples"] == W * H * 4
    assert s4["invalid_samples"] == 0
    assert abs(f4[..., :3].sum() / f2[..., :3].sum() - 2.0) < 0.05
    assert abs(s4["closest_rays"] / s2["closest_rays"] - 2.0) < 0.02
    # tile list in one pass == whole image
    ctx.clear_film(); ctx.render_tiles(0, 2, [(0, y, W, min(y + 32, H)) for y in range(0, H, 32)])
    ft = ctx.read_film()
    assert np.array_equal(ft[..., 3], f2[..., 3]) and np.allclose(ft, f2, rtol=1e-4, atol=1e-5)
    # what ran is the kernel the benchmark is quoted on: 64-byte quantised four-box nodes, six-byte stack entries ...
    assert s4["extend_width"] == 4 and s4["extend_node_bytes"] == 64 and ctx.trace_kernel_name() == "k_trace_pool4q"
    # ... and the same tree through the 128-byte float nodes gives the same film: quantised boxes only ever grow, the stack's entry
    # distances are only ever rounded down, so the same records are reached (pixels may differ where two triangles tie in t and the
    # children's order by entry distance differs)
    os.environ.update({"NORI_HIP_BVH": "device", "NORI_HIP_QUANT": "0"})
    try:
        hf = synthetic.make_c4(2236, W, H, 2)            # (a context of its own: the knobs are read by nori_hip_create)
        cf = hf.preprocess()
    finally:
        del os.environ["NORI_HIP_BVH"], os.environ["NORI_HIP_QUANT"]
    cf.clear_film(); cf.reset_stats(); cf.render(0, 2)
    ff, sf = cf.read_film(), cf.stats()
    assert sf["extend_node_bytes"] == 128
    assert np.array_equal(ff[..., 3], f2[..., 3])
    # (a render of this scene runs as two pipelines side by side, DESIGN.md finding 41: a pixel's samples reach the film in an order that
    # differs from run to run more than with one, so "the same pixel" is: the same sums up to the order of the additions)
    assert s2["pipelines"] == 2 and sf["pipelines"] == 2
    same = (ff == f2).all(-1).mean()
    close = np.isclose(ff, f2, rtol=1e-5, atol=1e-6).all(-1).mean()
    print("full-size C4, quantised against float nodes: %.5f of the pixels bit-identical, %.5f equal up to summation order" % (same, close))
    assert same > 0.98 and close > 0.995, (same, close)
    # ray by ray (nori_hip_trace walks the quantised nodes with cut stack entries in ctx, the two-box float nodes in cf): the same
    # distance for every ray, bit for bit, and another primitive only where two tie at that distance
    bounce = rays.copy(); bounce[:, 0:3] = full[:, 0:3]; bounce[:, 3] = 1e-4
    d = rng.normal(size=(len(rays), 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[(d * full[:, 8:11]).sum(1) < 0] *= -1; bounce[:, 4:7] = d; bounce[:, 7] = np.inf
    for r in (rays, bounce):
        a, b = ctx.trace(r), cf.trace(r)
        assert np.array_equal(a["t"], b["t"])
        ties = (a["prim"] != b["prim"]).mean()
        print("full-size C4, quantised against float nodes, %d rays: %.6f name another primitive at the same t" % (len(r), ties))
        assert ties < 0.005, ties
    sh2 = bounce.copy(); sh2[:, 7] = 0.7
    assert np.array_equal(ctx.trace(sh2, any_hit=True)["prim"], cf.trace(sh2, any_hit=True)["prim"])
    # rays that start behind their origin (mint < 0, as the probe rays of subsurface shapes do): the cut stack entries carry no sign,
    # so such rays keep full entries - same answers again
    neg = bounce.copy(); neg[:, 3] = -0.05; neg[:, 7] = 0.04
    a, b = ctx.trace(neg), cf.trace(neg)
    assert np.array_equal(a["t"], b["t"]) and (a["prim"] != b["prim"]).mean() < 0.005 and (a["t"][a["prim"] != NH.MISS] < 0.04 + 1e-6).all()
    assert ((a["prim"] != NH.MISS) & (a["t"] < 0)).any()          # (some of them do hit behind the origin)
    # directions with zero components (the general slab test of the quantised node step, Trav::box_step4q), also from points that lie
    # exactly on node planes (vertices of the height field)
    ax = np.zeros((6000, 8), np.float32); ax[:, 7] = np.inf; ax[:, 3] = 1e-4
    ax[:, 0:3] = full[:6000, 0:3] + [0, 0.3, 0]
    ax[:2000, 5] = -1; ax[2000:4000, 4] = 1; ax[4000:5000, 6] = -1; ax[5000:, 4] = np.float32(0.6); ax[5000:, 5] = np.float32(-0.8)
    a, b = ctx.trace(ax), cf.trace(ax)
    assert np.array_equal(a["t"], b["t"]) and (a["prim"] != b["prim"]).mean() < 0.005 and (a["prim"] != NH.MISS).mean() > 0.9
    assert abs(ff[..., :3].sum() - f2[..., :3].sum()) < 1e-4 * f2[..., :3].sum()
    hs.close(); hf.close()


@pytest.mark.skipif(os.environ.get("NORI_HIP_TEST_FULL") == "0", reason="NORI_HIP_TEST_FULL=0: skip the minute of host work (450 MB of OBJ text, a one-thread SAH build)")
def test_c4_full_size_hits_and_li_against_the_oracle(tmp_path):
    """The 10.0 M-triangle scene itself, not a reduced one: written as the reference's files, loaded by the oracle (its SAH tree, the
    reference's traversal order) and compared with the device-built tree on the GPU - closest hits the same primitive except on ties
    in t, any-hit answers identical, per-sample Li at matched streams."""
    W = H = 128
    xml = synthetic.write_c4(str(tmp_path), 2236, W, H, 1)
    o = O.OracleScene(xml, W, H, 1)
    os.environ["NORI_HIP_BVH"] = "device"
    try:
        hs = synthetic.make_c4(2236, W, H, 1)
        ctx = hs.preprocess()
    finally:
        del os.environ["NORI_HIP_BVH"]
    assert o.info()["prims"] == hs.info()["prims"] == 2 * 2236 * 2236 + 18
    rays, bounce = camera_and_bounce_rays(ctx, W, H, 50000, 8)
    assert np.array_equal(rays, o.camera_rays((np.random.default_rng(8).random((50000, 2)) * [W, H]).astype(np.float32)))
    for r in (rays, bounce):
        a, b = hits_as_u32(ctx.trace(r)), o.trace(r)
        same = a[:, 0] == b[:, 0]
        assert same.mean() > 0.995, same.mean()
        assert np.array_equal(a[same], b[same])                                   # prim, t, u, v bit for bit
        assert np.allclose(a[~same, 1].view(np.float32), b[~same, 1].view(np.float32), rtol=1e-5)     # a tie: the same distance
    sh = bounce.copy(); sh[:, 7] = 0.7
    assert np.array_equal(ctx.trace(sh, any_hit=True)["prim"], o.trace(sh, any_hit=True)[:, 0])
    n = 4096
    st = np.arange(n, dtype=np.uint64)
    li = ctx.li_samples(st, np.zeros(n, np.uint64), rays=rays[:n])
    oli = o.li_rays(rays[:n], st, np.zeros(n, np.uint64))
    err = np.abs(li - oli).max(1) / np.maximum(np.abs(oli).max(1), 1e-3)
    msg = "full-size C4: same primitive %.5f of bounce rays, Li within 1e-3 for %.4f (bit-identical %.4f) of %d samples, mean %.6f vs %.6f" % (
        same.mean(), (err < 1e-3).mean(), (li == oli).all(1).mean(), n, li.mean(), oli.mean())
    print(msg)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        open(os.path.join(out, "c4_full_size_parity.txt"), "w").write(msg + "\n")
    # device-built tree: another primitive can win a tie in t (a shared edge), and that sample's path differs from there on; all others
    # are the oracle's bit for bit (round 2: 0.9985 within 1e-3, 0.78 bit-identical)
    assert (li == oli).all(1).mean() > 0.995, (li == oli).all(1).mean()
    assert (err < 1e-3).mean() > 0.995, (err < 1e-3).mean()
    assert abs(li.mean() - oli.mean()) < 0.01 * oli.mean()
    hs.close()

