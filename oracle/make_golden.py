#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the REFERENCE's own compiled sources (oracle/_ref) and reports how the CPU
restatement (oracle/liboracle.so) compares.  Runs only in the build container (needs /root/reference).

    make -C oracle all _ref && python3 oracle/make_golden.py

Fixtures are data only: inputs (pixel samples, rays, seeds, BSDF queries) and the reference's outputs.
See oracle/ref/ref_capi.cpp for what is and is not reference code in _ref (the BVH build/traversal loop and the
re-seedable sampler wrapper are glue; bbox/triangle/sphere tests, hit records, camera, BSDFs, emitters, Li are not).
"""
import os
import re
import sys
import tempfile
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)
import oracle_lib as O  # noqa: E402
import ref_lib as R  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


class quiet:
    """silences the reference's cout chatter"""

    def __enter__(self):
        sys.stdout.flush()
        self.fd = os.dup(1)
        self.dn = os.open(os.devnull, os.O_WRONLY)
        os.dup2(self.dn, 1)

    def __exit__(self, *a):
        os.dup2(self.fd, 1)
        os.close(self.dn)
        os.close(self.fd)


def split_test_scenes(xml):
    """each <scene> of a <test> file as its own temporary scene XML (the reference runs tests inside the parser)"""
    s = open(xml).read()
    out = []
    for m in re.finditer(r"<scene>.*?</scene>", s, re.S):
        f = tempfile.NamedTemporaryFile("w", suffix=".xml", delete=False)
        f.write('<?xml version="1.0" encoding="utf-8"?>\n' + m.group(0))
        f.close()
        out.append(f.name)
    return out


def scene_fixture(name, xml, W, H, n_li, seed):
    print("== %s (%dx%d)" % (name, W, H))
    d = os.path.dirname(xml)
    o = O.OracleScene(xml, W, H, 1)
    with quiet():
        r = R.RefScene(R.override_xml(xml, W, H, 1), d)
    io, ir = o.info(), r.info()
    assert (io["shapes"], io["prims"], io["emitters"]) == (ir["shapes"], ir["prims"], ir["emitters"]), (io, ir)
    # meshes: OBJ loader parity
    for s in range(io["shapes"]):
        a = o.shape(s)
        if a["type"] != 0:
            continue
        b = r.mesh(s)
        for k in ("V", "F", "N", "UV"):
            if b[k] is None:
                assert a[k] is None
            else:
                assert np.array_equal(a[k], b[k]), "mesh %d %s differs" % (s, k)
    print("   meshes bit-identical (V, N, UV, F)")
    # primary rays: jittered, one per pixel
    rng = np.random.default_rng(seed)
    ys, xs = np.mgrid[0:H, 0:W]
    ps = np.stack([xs.ravel() + rng.random(W * H), ys.ravel() + rng.random(W * H)], 1).astype(np.float32)
    rays_ref = r.camera_rays(ps)
    rays_orc = o.camera_rays(ps)
    rel = np.abs(rays_ref[:, 4:7] - rays_orc[:, 4:7]).max()
    print("   camera rays: max |d_ref - d_oracle| = %.3g (matrix inverse differs in the last ulp, DESIGN.md)" % rel)
    assert rel < 2e-6
    hits_ref, full_ref = r.trace(rays_ref, full=True)
    hits_orc, full_orc = o.trace(rays_ref, full=True)
    print("   primary hits on the reference's rays: prim/t/uv identical = %s, records identical = %s, hit fraction %.3f" % (
        np.array_equal(hits_ref, hits_orc), np.array_equal(full_ref, full_orc), (hits_ref[:, 0] != 0xffffffff).mean()))
    # secondary rays: cosine-ish bounce rays from the hit points + shadow rays from a light sample to the hit points
    hit = hits_ref[:, 0] != 0xffffffff
    P = full_ref[hit, 0:3]
    Nn = full_ref[hit, 5:8]
    m = len(P)
    dirs = rng.normal(size=(m, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    flip = (dirs * Nn).sum(1) < 0
    dirs[flip] *= -1
    brays = np.zeros((m, 8), np.float32)
    brays[:, 0:3] = P
    brays[:, 3] = 1e-3
    brays[:, 4:7] = dirs
    brays[:, 7] = np.inf
    bh_ref, bf_ref = r.trace(brays, full=True)
    bh_orc, bf_orc = o.trace(brays, full=True)
    print("   bounce rays: identical = %s / %s" % (np.array_equal(bh_ref, bh_orc), np.array_equal(bf_ref, bf_orc)))
    # shadow rays between random pairs of hit points (any-hit)
    j = rng.permutation(m)
    seg = P[j] - P
    dist = np.linalg.norm(seg, axis=1).astype(np.float32)
    ok = dist > 1e-2
    srays = np.zeros((m, 8), np.float32)
    srays[:, 0:3] = P
    srays[:, 3] = 1e-3
    srays[:, 4:7] = seg / np.maximum(dist, 1e-9)[:, None]
    srays[:, 7] = dist - 1e-3
    srays = srays[ok]
    sh_ref = r.trace(srays, any_hit=True)
    sh_orc = o.trace(srays, any_hit=True)
    print("   shadow rays: identical = %s, occluded fraction %.3f" % (np.array_equal(sh_ref, sh_orc), (sh_ref[:, 0] == 1).mean()))
    # radiance per camera sample on the reference's rays, sampler seeded (i, 0)
    k = min(n_li, len(ps))
    sel = rng.permutation(len(ps))[:k]
    states = sel.astype(np.int64)
    seqs = np.zeros(k, np.int64)
    li_ref = r.li_rays(rays_ref[sel], states, seqs)
    li_orc = o.li_rays(rays_ref[sel], states, seqs)
    dd = np.abs(li_ref - li_orc).max(1)
    print("   Li per sample: bit-identical fraction %.4f, max abs diff %.3g, mean ref %s" % ((dd == 0).mean(), dd.max(), li_ref.mean(0)))
    np.savez_compressed(os.path.join(GOLD, name + ".npz"),
                        width=W, height=H, pixel_samples=ps, rays=rays_ref, hits=hits_ref, full=full_ref,
                        bounce_rays=brays, bounce_hits=bh_ref, bounce_full=bf_ref,
                        shadow_rays=srays, shadow_hits=sh_ref[:, 0].copy(),
                        li_index=sel.astype(np.int32), li=li_ref)
    return dict(hits=np.array_equal(hits_ref, hits_orc), li_exact=float((dd == 0).mean()))


def test_scene_fixtures():
    """per-sample radiance on the reference's own t-test scenes (1x1 pixel cameras), sampler seeded (i, 0)"""
    print("== per-sample Li on the t-test scenes")
    out = {}
    for fname in ("test-direct.xml", "test-furnace.xml"):
        xml = os.path.join(ROOT, "scenes", "tests", fname)
        ot = O.OracleScene(xml, is_test=True)
        for i, px in enumerate(split_test_scenes(xml)):
            if ot.test_meta(i)["integrator"] is None:
                continue
            with quiet():
                r = R.RefScene(px, os.path.dirname(xml))
            n = 4096
            rng = np.random.default_rng(100 + i)
            ps = rng.random((n, 2)).astype(np.float32)
            rays = r.camera_rays(ps)
            st = np.arange(n)
            li_ref = r.li_rays(rays, st, np.zeros(n, np.int64))
            li_orc = ot.li_rays(rays, st, np.zeros(n, np.int64), si=i)
            hits_ref = r.trace(rays)
            hits_orc = ot.trace(rays, si=i)
            key = "%s_%d" % (fname.replace(".xml", "").replace("-", "_"), i)
            dd = np.abs(li_ref - li_orc).max(1)
            print("   %-18s hits identical %s, Li bit-identical fraction %.4f (max diff %.3g), mean %.6f" % (
                key, np.array_equal(hits_ref, hits_orc), (dd == 0).mean(), dd.max(), li_ref.mean()))
            out[key + "_rays"] = rays
            out[key + "_hits"] = hits_ref
            out[key + "_li"] = li_ref
    np.savez_compressed(os.path.join(GOLD, "test_scenes.npz"), **out)


def bsdf_fixture():
    rng = np.random.default_rng(7)
    n = 2048
    wi = rng.normal(size=(n, 3)).astype(np.float32)
    wi /= np.linalg.norm(wi, axis=1, keepdims=True)
    wi = wi.astype(np.float32)
    smp = rng.random((n, 2)).astype(np.float32)
    cases = {
        "diffuse": (0, [0.725, 0.71, 0.68, 0.1, 1.5046, 1.000277]),
        "microfacet_default": (1, [0.5, 0.5, 0.5, 0.1, 1.5046, 1.000277]),
        "microfacet_kd0": (1, [0.0, 0.0, 0.0, 0.1, 1.5046, 1.000277]),
        "microfacet_rough": (1, [0.2, 0.25, 0.4, 0.3, 1.7, 1.0]),
        "dielectric_glass": (2, [0, 0, 0, 0.1, 1.5, 1.0]),
        "dielectric_water_glass": (2, [0, 0, 0, 0.1, 1.33, 1.5]),
        "mirror": (3, [0, 0, 0, 0.1, 1.5, 1.0]),
        # src/subsurface.cpp: both sample arguments carry `sample` (the model draws wo from the second one)
        "subsurface": (4, [0.9, 0.8, 0.7, 0.1, 1.3046, 1.000277]),
    }
    out = dict(wi=wi, sample=smp)
    print("== BSDF unit vectors")
    for k, (t, p) in cases.items():
        a = R.bsdf_sample(t, p, wi, smp)
        b = O.bsdf_sample(t, p, wi, smp)
        same = (a == b) | (np.isnan(a) & np.isnan(b))
        print("   %-24s identical fraction %.5f" % (k, same.all(1).mean()))
        out[k + "_type"] = t
        out[k + "_params"] = np.array(p, np.float32)
        out[k] = a
    # Subsurface::sampleDx alone: probe ray and disk sample for hits at p with shading normal nrm, sampler seeded (i, 0)
    p = rng.normal(size=(n, 3)).astype(np.float32)
    nrm = rng.normal(size=(n, 3)).astype(np.float32)
    nrm = (nrm / np.linalg.norm(nrm, axis=1, keepdims=True)).astype(np.float32)
    out["ss_p"] = p
    out["ss_n"] = nrm
    for tag, dm in (("ss_dx_0p1", [0.1, 0.1, 0.1]), ("ss_dx_0p01", [0.01, 0.01, 0.01]), ("ss_dx_4", [4, 4, 4])):
        a = R.ss_sample_dx(dm, p, nrm, np.arange(n), np.zeros(n))
        b = O.ss_sample_dx(dm, p, nrm, np.arange(n), np.zeros(n))
        print("   %-24s identical fraction %.5f" % (tag, (a == b).all(1).mean()))
        out[tag + "_dmfp"] = np.array(dm, np.float32)
        out[tag] = a
    np.savez_compressed(os.path.join(GOLD, "bsdf.npz"), **out)


def ttest_fixture():
    print("== t-test scenes (reference's test XMLs), 100000 paths, sampler seeded (scene index, 0)")
    res = {}
    for fname in ("test-direct.xml", "test-furnace.xml"):
        xml = os.path.join(ROOT, "scenes", "tests", fname)
        ot = O.OracleScene(xml, is_test=True)
        parts = split_test_scenes(xml)
        means = []
        for i, px in enumerate(parts):
            meta = ot.test_meta(i)
            if meta["integrator"] is None:
                means.append((np.nan, np.nan, meta["reference"]))
                continue
            with quiet():
                r = R.RefScene(px, os.path.dirname(xml))
            mr, vr = r.ttest(100000, i, 0)
            mo, vo = ot.test_run_one(i, i, 0, 100000)
            print("   %s[%d] reference value %.7g | ref mean %.9g var %.6g | oracle mean %.9g var %.6g | equal=%s" % (
                fname, i, meta["reference"], mr, vr, mo, vo, mr == mo and vr == vo))
            means.append((mr, vr, meta["reference"]))
        res[fname.replace(".xml", "").replace("-", "_")] = np.array(means)
    np.savez_compressed(os.path.join(GOLD, "ttest.npz"), **res)


def main():
    os.makedirs(GOLD, exist_ok=True)
    assert R.sampler_check() == 1, "glue sampler does not reproduce the reference's Independent sampler"
    kat = R.pcg32(42, 54, 6)
    assert [hex(x) for x in kat] == ["0xa15c02b7", "0x7b47f409", "0xba1d3330", "0x83d2f293", "0xbfa4784b", "0xcbed606e"], kat  # ext/pcg32/pcg32-demo.out:8
    assert np.array_equal(kat, O.pcg32(42, 54, 6))
    print("pcg32 KAT (ext/pcg32/pcg32-demo.out:8) ok for reference and oracle")
    S = os.path.join(ROOT, "scenes")
    only = sys.argv[1:]                      # fixture names to (re)generate; default all
    scenes = [("cbox", S + "/cbox/cbox_path_mis.xml", 64, 64, 2048, 1),
              ("cbox_mirror_microfacet", S + "/cbox/cbox_mirror_microfacet.xml", 64, 64, 2048, 2),
              ("table", S + "/table/table_path_mis.xml", 64, 48, 2048, 3),
              ("table_microfacet", S + "/table/table_path_mis_microfacet.xml", 64, 48, 2048, 4),
              # 105 k triangles, depth-26 tree, a 960-triangle mesh emitter (CDF search) and an emitter of zero radiance
              ("dragon", S + "/proj/dragon-diffuse.xml", 64, 48, 2048, 5),
              # two subsurface shapes (sphere mesh dmfp 0.1, cube dmfp 0.01): probe rays with the shape mask, sampleDx retries
              ("cbox_subsurface", S + "/proj/cbox_subsurface.xml", 64, 48, 4096, 6),
              # subsurface on the 100 k-triangle dragon (masked probes through a depth-26 tree)
              ("dragon_ss", S + "/proj/dragon-ss.xml", 64, 48, 2048, 8),
              # direct_mis (src/directMisIntegrator.cpp) on a subsurface slab with its own IORs
              ("odyssey_ss", S + "/proj/odyssey_ss.xml", 64, 48, 2048, 9),
              # the other path integrator of the reference's tests (src/pathMatsIntegrator.cpp): BSDF sampling only
              ("cbox_path_mats", S + "/cbox/cbox_path_mats.xml", 64, 64, 2048, 7),
              # the CPU renderer's procedural textures: checkerboard_color albedos, bump_map over checkerboard_float / constant_float
              # (src/checkerboard.cpp, src/consttexture.cpp, src/bumpMap.cpp: all compiled into _ref, no OpenCV needed)
              ("cbox_checker", S + "/cbox/cbox_checker.xml", 64, 64, 4096, 10)]
    for sc in scenes:
        if not only or sc[0] in only:
            scene_fixture(*sc)
    if not only or "test_scenes" in only:
        test_scene_fixtures()
    if not only or "bsdf" in only:
        bsdf_fixture()
    if not only or "ttest" in only:
        ttest_fixture()


if __name__ == "__main__":
    main()
