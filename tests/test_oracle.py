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
    "cbox_subsurface": "proj/cbox_subsurface.xml",       # subsurface shapes: probe rays with the shape mask (src/subsurface.cpp)
    "dragon_ss": "proj/dragon-ss.xml",                   # subsurface dragon: masked probes through the deep tree
    "odyssey_ss": "proj/odyssey_ss.xml",                 # direct_mis integrator on a subsurface slab with its own IORs
    "cbox_path_mats": "cbox/cbox_path_mats.xml",         # the other path integrator (src/pathMatsIntegrator.cpp)
    "cbox_checker": "cbox/cbox_checker.xml",             # checkerboard_color albedos, bump_map over checkerboard_float / constant_float (CPU plugins)
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
        if k.endswith("_type") or k.endswith("_params") or k in ("wi", "sample") or k.startswith("ss_"):
            continue
        out = O.bsdf_sample(int(g[k + "_type"]), g[k + "_params"], g["wi"], g["sample"])
        ref = g[k]
        assert np.array_equal(out, ref, equal_nan=True), k


def test_oracle_subsurface_probe_sampling_bit_exact():
    # Subsurface::sampleDx (src/subsurface.cpp:199-249) through the reference's own class: probe origin, direction,
    # mint, maxt and the disk sample, for three mean free paths
    g = np.load(os.path.join(GOLDEN, "bsdf.npz"))
    n = len(g["ss_p"])
    for tag in ("ss_dx_0p1", "ss_dx_0p01", "ss_dx_4"):
        out = O.ss_sample_dx(g[tag + "_dmfp"], g["ss_p"], g["ss_n"], np.arange(n), np.zeros(n))
        assert np.array_equal(out, g[tag]), tag


def _write_tiny_textured_scene(d, lookup):
    """a quad with uv coordinates and a 3x2 colour image (values chosen so that every texel differs)"""
    img = np.array([[[0, 10, 20], [50, 60, 70], [100, 110, 120]], [[150, 160, 170], [200, 210, 220], [250, 240, 230]]], np.uint8)
    with open(os.path.join(d, "t.ppm"), "wb") as f:
        f.write(b"P6\n# comment\n3 2\n255\n" + img.tobytes())
    with open(os.path.join(d, "g.pgm"), "wb") as f:
        f.write(b"P5\n3 2\n255\n" + img[:, :, 0].tobytes())
    with open(os.path.join(d, "q.obj"), "w") as f:
        f.write("v -1 -1 0\nv 1 -1 0\nv 1 1 0\nv -1 1 0\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nf 1/1 2/2 3/3 4/4\n")
    xml = '''<scene><integrator type="path_mis"/>
<camera type="perspective"><transform name="toWorld"><lookat origin="0,0,4" target="0,0,0" up="0,1,0"/></transform>
<integer name="width" value="8"/><integer name="height" value="8"/></camera>
<mesh type="obj"><string name="filename" value="q.obj"/>
 <normalModifier type="bump_map"><string name="filename" value="g.pgm"/><float name="height" value="50"/><string name="lookup" value="%s"/></normalModifier>
 <bsdf type="diffuse"><texture type="image_color" name="albedo"><string name="filename" value="t.ppm"/>
  <vector name="scale" value="0.5,2"/><point name="delta" value="0.25,0.125"/><string name="lookup" value="%s"/></texture></bsdf>
 <emitter type="area"><color name="radiance" value="1 1 1"/></emitter></mesh></scene>''' % (lookup, lookup)
    path = os.path.join(d, "scene_%s.xml" % lookup)
    with open(path, "w") as f:
        f.write(xml)
    return path, img


def test_oracle_image_lookups_on_a_known_image(tmp_path):
    # image order: the normal modifier is registered before the bsdf's texture (XML order)
    rng = np.random.default_rng(11)
    uv = rng.random((500, 2)).astype(np.float32) * 3 - 1
    path, img = _write_tiny_textured_scene(str(tmp_path), "nearest")
    o = O.OracleScene(path)
    ims = o.images()
    assert [(m["kind"], m["cols"], m["rows"], m["lookup"]) for m in ims] == [(1, 3, 2, 0), (0, 3, 2, 0)]
    assert np.array_equal(ims[1]["texels"].reshape(2, 3, 4)[:, :, :3], img) and np.all(ims[1]["texels"].reshape(2, 3, 4)[:, :, 3] == 255)
    # cv::Sobel 3x3, reflect-101 border, scale 50, written out by hand for the 2-row image (row -1 = row 1, col -1 = col 1)
    g = img[:, :, 0].astype(np.int64)
    pad = np.pad(g, 1, mode="reflect")
    gx = np.zeros((2, 3), np.int64)
    gy = np.zeros((2, 3), np.int64)
    for r in range(2):
        for c in range(3):
            w = pad[r:r + 3, c:c + 3]
            gx[r, c] = (w[0, 2] + 2 * w[1, 2] + w[2, 2]) - (w[0, 0] + 2 * w[1, 0] + w[2, 0])
            gy[r, c] = (w[2, 0] + 2 * w[2, 1] + w[2, 2]) - (w[0, 0] + 2 * w[0, 1] + w[0, 2])
    sob = np.frombuffer(ims[0]["texels"].tobytes(), np.int16).reshape(2, 3, 2)
    assert np.array_equal(sob[:, :, 0], np.clip(gx * 50, -32768, 32767)) and np.array_equal(sob[:, :, 1], np.clip(gy * 50, -32768, 32767))
    # nearest colour lookup = the CPU renderer's (src/imagetexture.cpp:49-77): x, y rescaled, row = x*rows, col = y*cols, value / 255
    x = uv[:, 0] / np.float32(0.5) - np.float32(0.25)
    y = uv[:, 1] / np.float32(2) - np.float32(0.125)
    x, y = x - np.floor(x), y - np.floor(y)
    row, col = np.minimum((x * 2).astype(int), 1), np.minimum((y * 3).astype(int), 2)
    assert np.array_equal(o.image_eval(1, uv), (img[row, col].astype(np.float32) / np.float32(255)))
    # bilinear = CUDA's tex2D(tex, y, x): texel centres at +0.5, wrap, 8-bit weights; then sRGB -> linear
    path, _ = _write_tiny_textured_scene(str(tmp_path), "bilinear")
    o2 = O.OracleScene(path)
    xb, yb = y.astype(np.float64) * 3 - 0.5, x.astype(np.float64) * 2 - 0.5
    i0, j0 = np.floor(xb).astype(int), np.floor(yb).astype(int)
    a, b = np.floor((xb - i0) * 256 + 0.5) / 256, np.floor((yb - j0) * 256 + 0.5) / 256
    t = img.astype(np.float64) / 255
    v = ((1 - a) * (1 - b))[:, None] * t[j0 % 2, i0 % 3] + (a * (1 - b))[:, None] * t[j0 % 2, (i0 + 1) % 3] + \
        ((1 - a) * b)[:, None] * t[(j0 + 1) % 2, i0 % 3] + (a * b)[:, None] * t[(j0 + 1) % 2, (i0 + 1) % 3]
    lin = np.where(v <= 0.04045, v / 12.92, ((v + 0.055) / 1.055) ** 2.4)
    assert np.abs(o2.image_eval(1, uv) - lin).max() < 2e-4            # float vs double, and weights that fall on a rounding boundary
    # bump: unit normals, equal to the frame normal where both gradients vanish
    nrm = np.tile(np.array([[0, 0, 1]], np.float32), (len(uv), 1))
    nb = o2.image_eval(0, uv, nrm)
    assert np.allclose(np.linalg.norm(nb, axis=1), 1, atol=1e-5)


def test_oracle_passes_the_references_own_ttests():
    # scenes/tests/test-direct.xml and test-furnace.xml: reference values from the XML, Sidak-corrected alpha as
    # ext/hypothesis/hypothesis.h:313-346
    from scipy import stats
    for fname in ("test-direct.xml", "test-furnace.xml"):
        o = O.OracleScene(os.path.join(SCENES, "tests", fname), is_test=True)
        n_tests = o.test_count()
        for i in range(n_tests):
            m = o.test_meta(i)
            if m["integrator"] is None:
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


def test_splat_film_mode_is_upstream_noris_filtered_put():
    """src/block.cpp:113-128 (dead in this fork behind the early return of :107-111): each sample goes to every pixel within
    the filter radius with weight w(dx) w(dy) from the 32-step table of the gaussian (radius 2, stddev 0.5); checked against
    a direct numpy evaluation for a handful of samples, and that weights stay a partition of the sample count in the mean"""
    o = O.OracleScene(os.path.join(SCENES, SCENE_XML["cbox"]), 24, 20, 1)
    table, radius = o.film_filter_table()
    assert radius == 2.0 and table[32] == 0 and abs(table[0] - (1 - np.exp(-8.0))) < 1e-6
    x = np.float32(2.0) * np.arange(32, dtype=np.float32) / np.float32(32)
    assert np.allclose(table[:32], np.maximum(0, np.exp(-2.0 * x * x) - np.exp(-8.0)), atol=1e-7)
    o.film_mode(True)
    o.render(0, 3, mode=1, threads=2)
    film = o.film()
    o2 = O.OracleScene(os.path.join(SCENES, SCENE_XML["cbox"]), 24, 20, 1)
    o2.render(0, 3, mode=1, threads=2)
    ref = o2.film()
    b = 2
    # same samples, same radiance: the splatted film is the reference film convolved with the separable table weights
    ys, xs = np.nonzero(ref[:, :, 3])
    assert ref[:, :, 3].sum() == 24 * 20 * 3
    # re-splat with numpy from the per-sample data of the oracle
    n = 24 * 20 * 3
    px = np.tile(np.repeat(np.arange(24), 1), 20 * 3)
    st = []
    li, ps = [], []
    for s in range(3):
        yy, xx = np.mgrid[0:20, 0:24]
        rgb, pos = o2.li_samples(xx.ravel(), yy.ravel(), (yy.ravel() * 24 + xx.ravel()).astype(np.uint64), np.full(480, s, np.uint64))
        li.append(rgb); ps.append(pos)
    li, ps = np.concatenate(li), np.concatenate(ps)
    want = np.zeros_like(film, dtype=np.float64)
    for (sx, sy), c in zip(ps, li):
        ox, oy = (int(sx) // 32) * 32, (int(sy) // 32) * 32
        fx, fy = np.float32(sx) - np.float32(0.5) - np.float32(ox - b), np.float32(sy) - np.float32(0.5) - np.float32(oy - b)
        for y in range(int(np.ceil(fy - 2)), int(np.floor(fy + 2)) + 1):
            wy = table[int(abs(np.float32(y) - fy) * np.float32(16))]
            for x in range(int(np.ceil(fx - 2)), int(np.floor(fx + 2)) + 1):
                wx = table[int(abs(np.float32(x) - fx) * np.float32(16))]
                want[y + oy, x + ox, :3] += c.astype(np.float64) * wx * wy
                want[y + oy, x + ox, 3] += float(wx) * float(wy)
    assert np.allclose(film, want, rtol=1e-4, atol=1e-5)
    assert film[:, :, 3].max() < ref[:, :, 3].max() * 1.5 and film[b + 3:-b - 3, b + 3:-b - 3, 3].min() > 0
