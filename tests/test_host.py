"""CPU: the host plugin surface (libnori_host.so) — loader, BVH build and flattening — against the oracle, plus the
C ABI's exported symbols.  No kernel is launched here."""
import ctypes as C
import os
import re
import shutil

import numpy as np
import pytest

import nori_hip as NH
import oracle_lib as O
from conftest import ROOT, SCENES

XMLS = ["cbox/cbox_path_mis.xml", "cbox/cbox_mirror_microfacet.xml", "table/table_path_mis.xml", "table/table_path_mis_microfacet.xml", "proj/dragon-diffuse.xml",
        "proj/cbox_subsurface.xml", "proj/textureAndBump.xml", "proj/normal.xml", "cbox/cbox_checker.xml"]


def test_c_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "nori_hip.h")).read()
    names = set(re.findall(r"\b(nori_hip_\w+)\s*\(", hdr))
    assert len(names) >= 20
    L = NH.hip_lib()
    for n in names:
        assert hasattr(L, n), n
    hdr = open(os.path.join(ROOT, "include", "nori_host.h")).read()
    L = NH.host_lib()
    for n in set(re.findall(r"\b(nori_host_\w+)\s*\(", hdr)):
        assert hasattr(L, n), n


def test_binding_structs_have_the_sizes_of_the_header(tmp_path):
    """The ctypes structures and numpy record types of nori_hip/__init__.py against sizeof() of the C structs they mirror, as a C
    compiler sees include/nori_hip.h (a field added on one side only - NoriHipStats.pipelines in round 3 - would shift every later one)."""
    import ctypes as C
    import subprocess
    pairs = [("NoriHipNode", NH.NODE_DTYPE.itemsize), ("NoriHipPrim", NH.PRIM_DTYPE.itemsize), ("NoriHipShadePrim", NH.SHADE_DTYPE.itemsize),
             ("NoriHipShape", NH.SHAPE_DTYPE.itemsize), ("NoriHipTexture", NH.TEXTURE_DTYPE.itemsize), ("NoriHipEmitter", NH.EMITTER_DTYPE.itemsize),
             ("NoriHipCamera", C.sizeof(NH.Camera)), ("NoriHipEnvMap", C.sizeof(NH.EnvMap)), ("NoriHipSceneDesc", C.sizeof(NH.SceneDesc)),
             ("NoriHipRay", NH.RAY_DTYPE.itemsize), ("NoriHipHit", NH.HIT_DTYPE.itemsize), ("NoriHipStats", C.sizeof(NH.Stats))]
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "nori_hip.h"\nint main(void) {\n' +
                   "".join('    printf("%s %%zu\\n", sizeof(%s));\n' % (n, n) for n, _ in pairs) + "    return 0;\n}\n")
    exe = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)])
    got = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    for name, size in pairs:
        assert int(got[name]) == size, (name, got[name], size)
    # and the last field of the statistics sits where the header puts it
    assert NH.Stats.pipelines.offset == C.sizeof(NH.Stats) - 8


@pytest.mark.parametrize("xml", XMLS)
def test_host_loader_and_bvh_equal_the_oracle(xml):
    path = os.path.join(SCENES, xml)
    hs = NH.HostScene(path, 96, 64, 2)
    o = O.OracleScene(path, 96, 64, 2)
    hi, oi = hs.info(), o.info()
    assert (hi["shapes"], hi["prims"], hi["emitters"], hi["width"], hi["height"], hi["nodes"]) == (oi["shapes"], oi["prims"], oi["emitters"], 96, 64, oi["nodes"])
    n32, idx = hs.bvh32()
    on, oidx = o.bvh()
    assert np.array_equal(np.frombuffer(n32.tobytes(), np.uint32).reshape(-1, 8), on)      # nodes bit-exact (flags, children, boxes)
    assert np.array_equal(idx, oidx)                                                       # leaf order bit-exact
    for s in range(hi["shapes"]):
        a, b = hs.mesh(s), o.shape(s)
        if a is None:                                  # analytic sphere
            assert b["type"] == 1
            continue
        for k in ("V", "F", "cdf"):
            assert np.array_equal(a[k], b[k])
        for k in ("N", "UV"):
            assert (a[k] is None) == (b[k] is None) and (a[k] is None or np.array_equal(a[k], b[k]))
    d = hs.tables()["desc"]
    s2c, c2w, _ = o.camera()
    assert np.array_equal(np.array(d.camera.sample_to_camera[:], np.float32).reshape(4, 4), s2c)
    assert np.array_equal(np.array(d.camera.camera_to_world[:], np.float32).reshape(4, 4), c2w)


def test_flattened_tables_are_consistent():
    hs = NH.HostScene(os.path.join(SCENES, XMLS[2]))
    T = hs.tables()
    n32, idx = hs.bvh32()
    nodes, prims, shade = T["nodes"], T["prims"], T["shade"]
    assert np.array_equal(prims["prim"], idx)                       # records are in m_indices order
    leaves = n32[(n32["w0"] & 1) == 1]
    inner = n32[(n32["w0"] & 1) == 0]
    assert len(nodes) == len(inner) == len(leaves) - 1
    last = (prims["shape_flags"] >> 31).astype(bool)
    assert last.sum() == len(leaves)
    ends = leaves["w1"] + (leaves["w0"] >> 1) - 1
    assert np.array_equal(np.sort(np.nonzero(last)[0]), np.sort(ends))
    # every child reference is in range, every leaf reference starts a leaf
    starts = set(int(x) for x in leaves["w1"])
    for side in ("left", "right"):
        r = nodes[side]
        isleaf = (r & 0x80000000) != 0
        assert np.all(r[~isleaf] < len(nodes))
        assert all(int(x & 0x7FFFFFFF) in starts for x in r[isleaf])
    # child boxes are the reference nodes' boxes: node k's left child is reference node (index of inner k) + 1
    inner_idx = np.nonzero((n32["w0"] & 1) == 0)[0]
    assert np.array_equal(nodes["lmin"], n32["bmin"][inner_idx + 1]) and np.array_equal(nodes["lmax"], n32["bmax"][inner_idx + 1])
    assert np.array_equal(nodes["rmin"], n32["bmin"][inner["w1"]]) and np.array_equal(nodes["rmax"], n32["bmax"][inner["w1"]])
    assert np.array_equal(prims["a"], shade["p0"]) and np.array_equal(prims["e1"], shade["p1"] - shade["p0"])


@pytest.mark.parametrize("xml", ["proj/textureAndBump.xml", "proj/normal.xml", "proj/cbox_subsurface.xml", "cbox/cbox_checker.xml"])
def test_images_and_subsurface_parameters_flatten_like_the_oracle(xml):
    """image tables (kind, size, uv transform, texels incl. the Sobel gradients of bump maps) and the per-shape
    texture / modifier / subsurface fields: host plugins vs the oracle's independent restatement"""
    path = os.path.join(SCENES, xml)
    hs, o = NH.HostScene(path, 64, 48, 1), O.OracleScene(path, 64, 48, 1)
    T = hs.tables()
    ims = o.images()
    # the host stores one table entry per plugin object, in shape order, albedo before modifier; map through the shapes
    assert len(T["textures"]) == len(ims)
    for s in range(hs.info()["shapes"]):
        osh, hsh = o.shape_images(s), T["shapes"][s]
        for field, k in (("albedo_tex", osh["albedo_tex"]), ("normal_tex", osh["normal_tex"])):
            assert (hsh[field] == 0) == (k < 0)
            if k < 0:
                continue
            t, m = T["textures"][hsh[field] - 1], ims[k]
            assert (t["kind"], t["lookup"], t["width"], t["height"]) == (m["kind"], m["lookup"], m["cols"], m["rows"])
            assert np.array_equal(t["scale"], m["scale"]) and np.array_equal(t["delta"], m["delta"])
            assert np.array_equal(T["tex_data"][int(t["offset"]):int(t["offset"]) + 4 * m["cols"] * m["rows"]], m["texels"])
        if hsh["bsdf"] == 4:
            assert hsh["alpha"] == np.sqrt(np.float32(3)) and hsh["ks"] == np.sqrt((osh["dmfp"] ** 2).sum(dtype=np.float32))   # ss_cs, ss_cd
            assert abs(hsh["int_ior"] - 1.3046) < 1e-6
    if "subsurface" in xml:
        assert sorted(T["shapes"]["bsdf"].tolist()).count(4) == 2 and len(ims) == 0


def test_texture_plugins_report_errors_like_the_reference(tmp_path):
    base = '<scene><integrator type="path_mis"/><camera type="perspective"/><mesh type="sphere">%s<emitter type="area"><color name="radiance" value="1 1 1"/></emitter></mesh></scene>'
    def load(body):
        p = tmp_path / "s.xml"
        p.write_text(base % body)
        return NH.HostScene(str(p))
    with pytest.raises(NH.NoriHipError, match="could not be found"):
        load('<bsdf type="diffuse"><texture type="image_color" name="albedo"><string name="filename" value="nope.jpg"/></texture></bsdf>')
    with pytest.raises(NH.NoriHipError, match="does not match any field"):
        load('<bsdf type="diffuse"><texture type="constant_color" name="kd"><color name="value" value="1 1 1"/></texture></bsdf>')
    with pytest.raises(NH.NoriHipError, match="already an albedo"):
        load('<bsdf type="subsurface"><color name="albedo" value="1 1 1"/><texture type="constant_color" name="albedo"/></bsdf>')
    (tmp_path / "g.pgm").write_bytes(b"P5\n2 2\n255\n" + bytes([0, 64, 128, 255]))
    mod = '<normalModifier type="bump_map"><string name="filename" value="g.pgm"/></normalModifier>'
    with pytest.raises(NH.NoriHipError, match="multiple Normal modifier"):
        load(mod + mod)
    with pytest.raises(NH.NoriHipError, match="not of size 2 or 3"):
        load('<normalModifier type="bump_map"><string name="filename" value="g.pgm"/><vector name="scale" value="1"/></normalModifier>')
    # the CPU renderer's bump_map takes a Texture<float> child (src/bumpMap.cpp:19-42); the texture kinds must fit their slot
    with pytest.raises(NH.NoriHipError, match="no texture with name bump_map"):
        load('<normalModifier type="bump_map"/>')
    with pytest.raises(NH.NoriHipError, match="does not match any field"):
        load('<normalModifier type="bump_map"><texture type="constant_float" name="height"/></normalModifier>')
    with pytest.raises(NH.NoriHipError, match="must be a float texture"):
        load('<normalModifier type="bump_map"><texture type="checkerboard_color" name="bump_map"/></normalModifier>')
    with pytest.raises(NH.NoriHipError, match="must be a colour texture"):
        load('<bsdf type="diffuse"><texture type="checkerboard_float" name="albedo"/></bsdf>')
    hs = load('<bsdf type="diffuse"><texture type="checkerboard_color" name="albedo"><color name="value2" value="0.25 0.5 0.75"/><vector name="scale" value="0.5,0.25"/></texture></bsdf>'
              '<normalModifier type="bump_map"><texture type="image_float" name="bump_map"><string name="filename" value="g.pgm"/></texture></normalModifier>')
    T = hs.tables()
    t0, t1 = T["textures"]
    assert (t0["kind"], t0["width"], t0["height"]) == (3, 6, 1) and t0["scale"].tolist() == [0.5, 0.25]
    assert T["tex_data"][int(t0["offset"]):int(t0["offset"]) + 24].view(np.float32).tolist() == [0, 0, 0, 0.25, 0.5, 0.75]     # value1 defaults to 0 (src/checkerboard.cpp:77)
    assert (t1["kind"], t1["width"], t1["height"]) == (5, 2, 2)
    assert T["tex_data"][int(t1["offset"]):int(t1["offset"]) + 16].reshape(4, 4)[:, 0].tolist() == [0, 64, 128, 255]
    o = O.OracleScene(str(tmp_path / "s.xml"))
    ims = o.images()
    assert [m["kind"] for m in ims] == [3, 5] and np.array_equal(ims[1]["texels"], T["tex_data"][int(t1["offset"]):int(t1["offset"]) + 16])
    # ImageTexture<float>::eval (src/imagetexture.cpp:26-46): row = x * rows, col = y * cols, value / 255 -> the normal (v, v, v)
    got = o.image_eval(1, np.array([[0.1, 0.1], [0.1, 0.6], [0.6, 0.1], [0.9, 0.9], [1.75, -0.25]], np.float32), np.zeros((5, 3), np.float32) + [0, 0, 1])
    assert np.array_equal(got[:, 0], (np.array([0, 64, 128, 255, 255], np.float32) / np.float32(255))) and np.all(got[:, 0:1] == got)
    hs = load(mod.replace("</normalModifier>", '<vector name="scale" value="2,4"/><point name="delta" value="0.5,0.25"/></normalModifier>'))
    t = hs.tables()["textures"][0]
    assert (t["kind"], t["width"], t["height"]) == (1, 2, 2) and t["scale"].tolist() == [2.0, 4.0] and t["delta"].tolist() == [0.5, 0.25]


def test_single_leaf_scene_and_test_files_load():
    ht = NH.HostScene(os.path.join(SCENES, "tests", "test-direct.xml"))
    assert ht.is_test and ht.scene_count == 10
    assert ht.test_meta(0)["path_mis"] and ht.test_meta(5)["path_mis"]          # rows 0-4 path_mats, 5-9 path_mis: both have the GPU backend
    assert abs(ht.test_meta(5)["reference"] - 0.0898394) < 1e-7
    T = ht.tables(5)
    assert T["desc"].root == 0x80000000 and len(T["nodes"]) == 0 and len(T["prims"]) == 3
    ot = O.OracleScene(os.path.join(SCENES, "tests", "test-direct.xml"), is_test=True)
    for i in range(10):
        n32, idx = ht.bvh32(i)
        on, oi = ot.bvh(i)
        assert np.array_equal(np.frombuffer(n32.tobytes(), np.uint32).reshape(-1, 8), on) and np.array_equal(idx, oi)


def test_parser_errors_carry_file_context():
    import tempfile
    bad = tempfile.NamedTemporaryFile("w", suffix=".xml", delete=False)
    bad.write("<scene>\n  <integrator type=\"path_mis\"/>\n  <bogus/>\n</scene>\n")
    bad.close()
    with pytest.raises(NH.NoriHipError, match=r'unexpected tag "bogus".*row 3'):
        NH.HostScene(bad.name)
    bad = tempfile.NamedTemporaryFile("w", suffix=".xml", delete=False)
    bad.write("<scene>\n  <integrator type=\"path_mis\"/>\n</scene>\n")
    bad.close()
    with pytest.raises(NH.NoriHipError, match="No camera was specified"):
        NH.HostScene(bad.name)
    bad = tempfile.NamedTemporaryFile("w", suffix=".xml", delete=False)
    bad.write("<scene><integrator type=\"nope\"/></scene>")
    bad.close()
    with pytest.raises(NH.NoriHipError, match='constructor for class "nope"'):
        NH.HostScene(bad.name)


def test_procedural_scene_builder_and_parallel_bvh_build_is_deterministic():
    rng = np.random.default_rng(3)
    n = 300
    xs, ys = np.meshgrid(np.linspace(-1, 1, n), np.linspace(-1, 1, n))
    V = np.stack([xs.ravel(), rng.random(n * n) * 0.1, ys.ravel()], 1).astype(np.float32)
    i = np.arange(n * n).reshape(n, n)
    a, b, c, d = i[:-1, :-1].ravel(), i[:-1, 1:].ravel(), i[1:, 1:].ravel(), i[1:, :-1].ravel()
    F = np.concatenate([np.stack([a, b, c], 1), np.stack([a, c, d], 1)]).astype(np.uint32)
    Lv = np.array([[-0.3, 1.5, -0.3], [0.3, 1.5, -0.3], [0.3, 1.5, 0.3], [-0.3, 1.5, 0.3]], np.float32)
    Lf = np.array([[0, 1, 2], [0, 2, 3]], np.uint32)
    res = []
    for _ in range(2):
        sb = NH.SceneBuilder()
        sb.add_mesh("field", V, F, color=(0.5, 0.5, 0.5))
        sb.add_mesh("light", Lv, Lf, radiance=(10, 10, 10))
        sb.set_camera((0, 1, 3), (0, 0, 0), (0, 1, 0), 40.0, 64, 64)
        hs = sb.finish(spp=2)
        n32, idx = hs.bvh32()
        res.append((n32.copy(), idx.copy()))
        assert hs.info()["prims"] == len(F) + 2
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])


def test_obj_reader_equals_the_reference_loader_on_awkward_files(tmp_path):
    """host/objreader.cpp (mapped file, parallel pieces, vertex numbers by first use) against the oracle's loader, which is
    bit-identical to the reference's (src/obj.cpp:32-132, oracle/make_golden.py): quads, every face-token form, CRLF, tabs, signs,
    exponents, numbers that need more than a double to round correctly, a file long enough to be cut into several pieces."""
    rng = np.random.default_rng(17)
    nv = 60000
    fmts = ["%.9g", "%.6f", "%.17g", "%e", "%.3f", "%+.8g", "%.12e", "%.25f"]
    rows, want = [], []
    for i in range(nv):
        v = rng.normal(size=3) * 10.0 ** rng.integers(-6, 6)
        if i % 1000 == 0:                       # halfway cases between two floats: the text is a float midpoint, exactly or off by one digit in the 40th place
            f0 = np.float32(v[0]); mid = (np.float64(f0) + np.float64(np.nextafter(f0, np.float32(np.inf)))) / 2
            txt = ["%.60g" % mid, "%.60g1" % mid if "." in "%.60g" % mid else "%.60g.0001" % mid, fmts[i % len(fmts)] % v[2]]
        else:
            txt = [fmts[(i + k) % len(fmts)] % v[k] for k in range(3)]
        rows.append("v" + ("\t" if i % 7 == 0 else " ") + "  ".join(txt) + ("\r" if i % 3 == 0 else ""))
        want.append(txt)
    lines = ["# awkward", "o thing", "s off"] + rows
    lines += ["vt %.7g %.7g" % tuple(rng.random(2)) for _ in range(nv)] + ["vn %.7g %.7g %.7g" % tuple(rng.normal(size=3)) for _ in range(nv)]
    faces = []
    for k in range(90000):
        a, b, c, d = (int(x) + 1 for x in rng.integers(0, nv, 4))
        form = k % 4
        tok = (lambda i: "%d/%d/%d" % (i, (i * 7) % nv + 1, i)) if form == 0 else (lambda i: "%d/%d/%d" % (i, i, (i * 3) % nv + 1)) if form == 1 else (lambda i: "%d/%d/%d" % (i, i, i))
        faces.append("f " + " ".join(tok(i) for i in ((a, b, c, d) if k % 5 == 0 else (a, b, c))) + ("  " if k % 11 == 0 else ""))
    lines += faces
    (tmp_path / "m.obj").write_text("\n".join(lines))                      # (no newline at the end of the file)
    xml = tmp_path / "s.xml"
    xml.write_text('<scene><integrator type="path_mis"/><camera type="perspective"/><mesh type="obj"><string name="filename" value="m.obj"/>'
                   '<transform name="toWorld"><scale value="1,2,0.5"/><translate value="0.25,0,0"/></transform>'
                   '<emitter type="area"><color name="radiance" value="1 1 1"/></emitter></mesh></scene>')
    os.environ["NORI_HIP_BVH"] = "device"                                 # no host tree: this is about the loader
    try:
        hs = NH.HostScene(str(xml), 16, 16, 1)
    finally:
        del os.environ["NORI_HIP_BVH"]
    o = O.OracleScene(str(xml), 16, 16, 1)
    a, b = hs.mesh(0), o.shape(0)
    assert len(a["F"]) == 90000 + 18000
    for k in ("V", "N", "UV", "F"):
        assert np.array_equal(a[k], b[k]), k
    # the positions-only form takes the direct table: the same file without vt / vn and with plain indices
    lines2 = rows + ["f %d %d %d" % tuple(int(x) + 1 for x in rng.integers(0, nv, 3)) for _ in range(50000)]
    (tmp_path / "m.obj").write_text("\n".join(lines2) + "\n")
    os.environ["NORI_HIP_BVH"] = "device"
    try:
        hs2 = NH.HostScene(str(xml), 16, 16, 1)
    finally:
        del os.environ["NORI_HIP_BVH"]
    o2 = O.OracleScene(str(xml), 16, 16, 1)
    a, b = hs2.mesh(0), o2.shape(0)
    assert a["N"] is None and a["UV"] is None and np.array_equal(a["V"], b["V"]) and np.array_equal(a["F"], b["F"])
    # number parsing alone: every text, untransformed, is the float numpy (strtod then one rounding is NOT what it does: it parses to float32 correctly) gives
    (tmp_path / "s.xml").write_text('<scene><integrator type="path_mis"/><camera type="perspective"/><mesh type="obj"><string name="filename" value="m.obj"/>'
                                    '<emitter type="area"><color name="radiance" value="1 1 1"/></emitter></mesh></scene>')
    (tmp_path / "m.obj").write_text("\n".join(rows + ["f %d %d %d" % (i + 1, i + 2, i + 3) for i in range(0, nv - 2, 3)]) + "\n")
    os.environ["NORI_HIP_BVH"] = "device"
    try:
        hs3 = NH.HostScene(str(xml), 16, 16, 1)
    finally:
        del os.environ["NORI_HIP_BVH"]
    V = hs3.mesh(0)["V"]
    import ctypes
    libc = ctypes.CDLL("libc.so.6"); libc.strtof.restype = ctypes.c_float; libc.strtof.argtypes = [ctypes.c_char_p, ctypes.c_void_p]
    ref = np.array([[libc.strtof(t.encode(), None) for t in row] for row in want[:len(V)]], np.float32)
    assert np.array_equal(V, ref)
    with pytest.raises(NH.NoriHipError, match="names position"):
        (tmp_path / "m.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 4\n")
        NH.HostScene(str(xml), 16, 16, 1)
    with pytest.raises(NH.NoriHipError, match="Invalid vertex data"):
        (tmp_path / "m.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1/1/1/1 2 3\n")
        NH.HostScene(str(xml), 16, 16, 1)
    with pytest.raises(NH.NoriHipError, match="Could not parse integer"):
        (tmp_path / "m.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 -2 3\n")
        NH.HostScene(str(xml), 16, 16, 1)


def test_binary_mesh_cache_round_trip(tmp_path):
    """NORI_HIP_MESH_CACHE=1: the second load reads "<obj>.noricache" instead of parsing the text and
    yields the same arrays, BVH and tables; a cache made from another toWorld or an older OBJ is ignored."""
    import shutil
    d = tmp_path / "scene"
    shutil.copytree(os.path.join(SCENES, "cbox"), d)
    xml = str(d / "cbox_path_mis.xml")
    plain = NH.HostScene(xml, 32, 32, 1)
    os.environ["NORI_HIP_MESH_CACHE"] = "1"
    try:
        first = NH.HostScene(xml, 32, 32, 1)                       # parses, writes the caches
        caches = sorted(p.name for p in (d / "meshes").glob("*.noricache"))
        assert len(caches) == 4, caches
        # replace the OBJ text behind one cache entry by comments of the same size and restore its mtime: a load that
        # parsed the text would now see an empty mesh, a cached load cannot notice
        victim = d / "meshes" / caches[0].replace(".noricache", "")
        st = os.stat(victim)
        data = victim.read_bytes()
        victim.write_bytes(b"#" * len(data))
        os.utime(victim, ns=(st.st_atime_ns, st.st_mtime_ns))
        second = NH.HostScene(xml, 32, 32, 1)                      # reads the caches
        # and a changed file (new mtime) is parsed again
        victim.write_bytes(data)
        os.utime(victim, ns=(st.st_atime_ns, st.st_mtime_ns + 10**9))
        third = NH.HostScene(xml, 32, 32, 1)
    finally:
        del os.environ["NORI_HIP_MESH_CACHE"]
    for hs in (first, second, third):
        assert hs.info() == plain.info()
        for s in range(plain.info()["shapes"]):
            a, b = plain.mesh(s), hs.mesh(s)
            for k in ("V", "F", "cdf", "N", "UV"):
                assert (a[k] is None) == (b[k] is None) and (a[k] is None or np.array_equal(a[k], b[k])), (s, k)
        n0, i0 = plain.bvh32()
        n1, i1 = hs.bvh32()
        assert np.array_equal(n0, n1) and np.array_equal(i0, i1)


def test_host_tree_cache_round_trip(tmp_path):
    """$NORI_HIP_BVH_CACHE: the reference-order SAH tree is written once per distinct set of primitive boxes and read back
    afterwards; a changed mesh has another key, a damaged entry is ignored"""
    d = str(tmp_path / "scene"); cache = tmp_path / "cache"; cache.mkdir()
    shutil.copytree(os.path.join(SCENES, "cbox"), d)
    xml = os.path.join(d, "cbox_path_mis.xml")
    ref_nodes, ref_idx = NH.HostScene(xml, 32, 32, 1).bvh32()
    ref_nodes, ref_idx = ref_nodes.copy(), ref_idx.copy()
    os.environ["NORI_HIP_BVH_CACHE"] = str(cache)
    try:
        for threads in ("3", "8"):                              # the key does not depend on how many threads hash and build
            os.environ["NORI_HIP_LOAD_THREADS"] = threads
            hs = NH.HostScene(xml, 32, 32, 1)
            n, i = hs.bvh32()
            assert np.array_equal(n, ref_nodes) and np.array_equal(i, ref_idx)
            assert len(list(cache.iterdir())) == 1
            hs.close()
        entry = next(cache.iterdir())
        raw = bytearray(entry.read_bytes())
        entry.write_bytes(bytes(raw[:len(raw) // 2]))                       # truncated entry: rebuilt, and rewritten whole
        hs = NH.HostScene(xml, 32, 32, 1)
        n, i = hs.bvh32()
        assert np.array_equal(n, ref_nodes) and np.array_equal(i, ref_idx) and entry.stat().st_size == len(raw)
        raw[-4:] = (0xFFFFFFF0).to_bytes(4, "little")                       # an index out of range: refused
        entry.write_bytes(bytes(raw))
        n, i = NH.HostScene(xml, 32, 32, 1).bvh32()
        assert np.array_equal(i, ref_idx)
        # another mesh: another entry
        obj = os.path.join(d, "meshes", "light.obj")
        open(obj, "a").write("\nv 0 1.5 0\nv 0.1 1.5 0\nv 0 1.5 0.1\nf 5/1 6/1 7/1\n")
        NH.HostScene(xml, 32, 32, 1)
        assert len(list(cache.iterdir())) == 2
    finally:
        del os.environ["NORI_HIP_BVH_CACHE"]
        os.environ.pop("NORI_HIP_LOAD_THREADS", None)


def test_png_and_pfm_writers(tmp_path):
    """the headless driver's outputs: PFM (raw floats) and the 8-bit sRGB PNG of Bitmap::savePNG (src/bitmap.cpp:109-123)"""
    from PIL import Image
    rng = np.random.default_rng(2)
    img = (rng.random((37, 301, 3)) ** 2 * 1.2).astype(np.float32)          # some values above 1: clamped; 301*3+1 bytes per row
    img[0, 0] = [np.nan, -1.0, 0.5]
    NH.write_image(tmp_path / "a.png", img)
    got = np.asarray(Image.open(tmp_path / "a.png").convert("RGB"))
    v = np.where(img <= 0.0031308, 12.92 * img, 1.055 * np.power(np.maximum(img, 0), 1 / 2.4) - 0.055) * 255
    want = np.where(np.isnan(v) | (v < 0), 0, np.minimum(v, 255)).astype(np.uint8)
    assert got.shape == want.shape and np.abs(got.astype(int) - want.astype(int)).max() <= 1
    big = np.tile(img, (8, 1, 1))                                            # > 65535 bytes: several stored deflate blocks
    NH.write_image(tmp_path / "b.png", big)
    assert np.asarray(Image.open(tmp_path / "b.png")).shape == big.shape
    NH.write_image(tmp_path / "a.pfm", img)
    raw = (tmp_path / "a.pfm").read_bytes()
    assert raw.startswith(b"PF\n301 37\n-1.0\n")
    back = np.frombuffer(raw[len(b"PF\n301 37\n-1.0\n"):], np.float32).reshape(37, 301, 3)[::-1]
    assert np.array_equal(back, img, equal_nan=True)


def test_exr_writer_follows_the_openexr_layout(tmp_path):
    """the reference's native output (src/bitmap.cpp:82-107).  Read back here by walking the file as the OpenEXR file-layout
    document describes it: magic, version, attributes, line-offset table, scanline blocks with channels in alphabetical order"""
    import struct
    rng = np.random.default_rng(6)
    img = rng.random((19, 33, 3)).astype(np.float32) * 4
    NH.write_image(tmp_path / "a.exr", img)
    b = (tmp_path / "a.exr").read_bytes()
    assert struct.unpack_from("<iI", b, 0) == (20000630, 2)
    pos, attrs = 8, {}
    def cstr(p):
        e = b.index(b"\0", p)
        return b[p:e].decode(), e + 1
    while b[pos] != 0:
        name, pos = cstr(pos)
        typ, pos = cstr(pos)
        size, = struct.unpack_from("<i", b, pos)
        attrs[name] = (typ, b[pos + 4:pos + 4 + size])
        pos += 4 + size
    pos += 1
    assert {"channels", "compression", "dataWindow", "displayWindow", "lineOrder", "pixelAspectRatio", "screenWindowCenter", "screenWindowWidth"} <= set(attrs)
    assert attrs["compression"] == ("compression", b"\0") and attrs["lineOrder"] == ("lineOrder", b"\0")
    assert struct.unpack("<4i", attrs["dataWindow"][1]) == (0, 0, 32, 18)
    ch, p, names = attrs["channels"][1], 0, []
    while ch[p] != 0:
        e = ch.index(b"\0", p)
        names.append(ch[p:e].decode())
        assert struct.unpack_from("<iB3xii", ch, e + 1) == (2, 0, 1, 1)      # FLOAT, sampling 1 x 1
        p = e + 1 + 16
    assert names == ["B", "G", "R"]
    offsets = struct.unpack_from("<19Q", b, pos)
    out = np.zeros_like(img)
    for y, off in enumerate(offsets):
        yy, nbytes = struct.unpack_from("<ii", b, off)
        assert (yy, nbytes) == (y, 33 * 12)
        planes = np.frombuffer(b, np.float32, 33 * 3, off + 8).reshape(3, 33)
        out[y] = planes[::-1].T                                               # B, G, R planes -> r, g, b
    assert off + 8 + 33 * 12 == len(b)
    assert np.array_equal(out, img)


def test_synthetic_scene_is_closed_and_lit_from_above(tmp_path):
    """nori_hip/synthetic.py (BASELINE config 4 at reduced size): walls face inwards, the four light quads face DOWN - area
    emitters are one-sided (src/arealight.cpp:41-54); wound the other way round they light only the ceiling above them, which is
    what the generator did until round 2 - and the oracle's emitter samples from the floor carry radiance."""
    from nori_hip import synthetic
    meshes = synthetic.c4_meshes(16)
    assert sum(len(F) for _, _, F, _ in meshes) == 2 * 16 * 16 + 18
    for name, V, F, rad in meshes:
        n = np.cross(V[F[:, 1]] - V[F[:, 0]], V[F[:, 2]] - V[F[:, 0]])
        c = V[F].mean(1)
        if name == "heightfield":
            assert (n[:, 1] > 0).all()                                      # the terrain faces up
        elif rad is not None:
            assert (n[:, 1] < 0).all() and (n[:, [0, 2]] == 0).all()        # lights face the terrain
        else:
            assert ((n * (np.array([0, 1, 0], np.float32) - c)).sum(1) > 0).all(), name     # walls and ceiling face the inside
    xml = synthetic.write_c4(str(tmp_path), 16, 32, 32, 1)
    o = O.OracleScene(xml, 32, 32, 1)
    ys, xs = np.mgrid[0:32, 0:32]
    rays = o.camera_rays(np.stack([xs.ravel() + 0.5, ys.ravel() + 0.5], 1).astype(np.float32))
    n_ = len(rays)
    li = o.li_rays(rays, np.arange(n_, dtype=np.uint64), np.zeros(n_, np.uint64))
    assert (li.max(1) > 0).mean() > 0.5          # most one-sample paths see a light directly (NEE), not only through the ceiling
