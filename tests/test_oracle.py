"""CPU: the oracle (oracle/liboracle.so) against the fixtures generated from the reference's own compiled sources
(tests/golden/*.npz, made by oracle/make_golden.py), the reference's published golden vectors, and the statistics
the reference build printed during the survey."""
import os

import numpy as np
import pytest

import oracle_lib as O
from conftest import GOLDEN, SCENES

SCENE_XML = {
    "cbox": "cbox/cbox_path_mis.xml",
    "cbox_mirror_microfacet": "cbox/cbox_mirror_microfacet.xml",
    "table": "table/table_path_mis.xml",
    "table_microfacet": "table/table_path_mis_microfacet.xml",
    "dragon": "proj/dragon-diffuse.xml",
}


def test_pcg32_known_answer():
    # ext/pcg32/pcg32-demo.out:8 (seed 42, 54)
    assert [hex(x) for x in O.pcg32(42, 54, 6)] == ["0xa15c02b7", "0x7b47f409", "0xba1d3330", "0x83d2f293", "0xbfa4784b", "0xcbed606e"]


def test_bvh_statistics_match_reference_build():
    # numbers printed by the reference binary (1 TBB thread)
    o = O.OracleScene(os.path.join(SCENES, SCENE_XML["cbox"]))
    st = o.bvh_stats()
    assert (st["nodes"], st["leaves"], st["max_leaf"], st["depth"]) == (7065, 3533, 6, 17)
    assert abs(st["sah"] - 36.92) < 0.005
    o = O.OracleScene(os.path.join(SCENES, SCENE_XML["table"]))
    st = o.bvh_stats()
    assert (st["nodes"], st["max_leaf"], st["depth"]) == (13643, 14, 21)


@pytest.mark.parametrize("name", list(SCENE_XML))
def test_oracle_reproduces_reference_bit_for_bit(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    W, H = int(g["width"]), int(g["height"])
    o = O.OracleScene(os.path.join(SCENES, SCENE_XML[name]), W, H, 1)
    # camera: matrix inverse differs from Eigen's SSE kernel in the last ulp (DESIGN.md "float parity")
    assert np.abs(o.camera_rays(g["pixel_samples"])[:, 4:7] - g["rays"][:, 4:7]).max() < 2e-6
    hits, full = o.trace(g["rays"], full=True)
    assert np.array_equal(hits, g["hits"])            # primitive id, t, uv: bit-exact
    assert np.array_equal(full, g["full"])            # p, uv, shading / geometric normal, shape
    bh, bf = o.trace(g["bounce_rays"], full=True)
    assert np.array_equal(bh, g["bounce_hits"]) and np.array_equal(bf, g["bounce_full"])
    sh = o.trace(g["shadow_rays"], any_hit=True)
    assert np.array_equal(sh[:, 0], g["shadow_hits"])
    sel = g["li_index"]
    li = o.li_rays(g["rays"][sel], sel.astype(np.uint64), np.zeros(len(sel), np.uint64))
    assert np.array_equal(li, g["li"])                # PathMisIntegrator::Li per camera sample: bit-exact


def test_oracle_li_on_reference_test_scenes():
    g = np.load(os.path.join(GOLDEN, "test_scenes.npz"))
    for fname in ("test-direct.xml", "test-furnace.xml"):
        o = O.OracleScene(os.path.join(SCENES, "tests", fname), is_test=True)
        for i in range(o.test_count()):
            key = "%s_%d" % (fname.replace(".xml", "").replace("-", "_"), i)
            if key + "_rays" not in g:
                continue
            rays = g[key + "_rays"]
            n = len(rays)
            assert np.array_equal(o.trace(rays, si=i), g[key + "_hits"])
            assert np.array_equal(o.li_rays(rays, np.arange(n), np.zeros(n), si=i), g[key + "_li"])


def test_oracle_bsdfs_bit_exact():
    g = np.load(os.path.join(GOLDEN, "bsdf.npz"))
    for k in g.files:
        if k.endswith("_type") or k.endswith("_params") or k in ("wi", "sample"):
            continue
        out = O.bsdf_sample(int(g[k + "_type"]), g[k + "_params"], g["wi"], g["sample"])
        ref = g[k]
        assert np.array_equal(out, ref, equal_nan=True), k


def test_oracle_passes_the_references_own_ttests():
    # scenes/tests/test-direct.xml and test-furnace.xml: reference values from the XML, Sidak-corrected alpha as
    # ext/hypothesis/hypothesis.h:313-346
    from scipy import stats
    for fname in ("test-direct.xml", "test-furnace.xml"):
        o = O.OracleScene(os.path.join(SCENES, "tests", fname), is_test=True)
        n_tests = o.test_count()
        for i in range(n_tests):
            m = o.test_meta(i)
            if not m["path_mis"]:
                continue
            mean, var = o.test_run_one(i, i, 0, 100000 if "direct" in fname else 30000)
            n = 100000 if "direct" in fname else 30000
            t = abs(mean - m["reference"]) * np.sqrt(n / max(var, 1e-5))
            p = 2 * (1 - stats.t.cdf(t, n - 1))
            alpha = 1 - (1 - m["significance"]) ** (1.0 / n_tests)
            assert p >= alpha, (fname, i, mean, m["reference"], p)


def test_film_put_is_the_half_pixel_shifted_box():
    # a sample at x+u lands on pixel x-1 when u < 0.5 and on x otherwise; weight +1; toBitmap = rgb / w
    W, H = 40, 36
    pos = np.array([[5.25, 7.75], [5.75, 7.25], [0.2, 0.2], [39.9, 35.9], [33.49, 33.5]], np.float32)
    rgb = np.ones((5, 3), np.float32)
    f = O.film_put(W, H, 2.0, pos, rgb)
    b = 2
    expect = [(4, 7), (5, 6), (-1, -1), (39, 35), (32, 33)]
    got = np.argwhere(f[:, :, 3] > 0)
    assert sorted((int(x) - b, int(y) - b) for y, x in got) == sorted(expect)
    # invalid radiance is dropped (src/block.cpp:94-98)
    f = O.film_put(W, H, 2.0, pos[:2], np.array([[np.nan, 0, 0], [-1, 0, 0]], np.float32))
    assert f.sum() == 0


def test_cbox_image_mean_matches_the_reference_render():
    # the reference binary's cbox 512x512 at 64 spp has mean RGB 0.150194 / 0.116927 / 0.125327.
    # 8 spp in the reference's block-stream mode is enough to land within a few standard errors of it.
    o = O.OracleScene(os.path.join(SCENES, SCENE_XML["cbox"]), 512, 512, 8)
    o.render(0, 8, mode=0, threads=8)
    mean = o.bitmap().reshape(-1, 3).mean(0, dtype=np.float64)
    assert np.all(np.abs(mean - [0.150194, 0.116927, 0.125327]) < 1.5e-3), mean
