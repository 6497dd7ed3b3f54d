"""ctypes bindings for the CPU oracle (oracle/liboracle.so) — TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(ROOT, "oracle", "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/liboracle.so missing: run `make -C oracle` (or __graft_entry__.build())")
        L = C.CDLL(path)
        L.orc_last_error.restype = C.c_char_p
        L.orc_load_scene.restype = C.c_void_p
        L.orc_load_scene.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int]
        L.orc_load_test.restype = C.c_void_p
        L.orc_load_test.argtypes = [C.c_char_p]
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_test_reference.restype = C.c_float
        L.orc_test_significance.restype = C.c_float
        L.orc_render.restype = C.c_double
        L.orc_render.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class OracleScene:
    """A scene (or a <test type=ttest> suite when is_test) loaded by the oracle."""

    def __init__(self, xml, width=0, height=0, spp=0, is_test=False):
        L = lib()
        self.is_test = is_test
        h = L.orc_load_test(xml.encode()) if is_test else L.orc_load_scene(xml.encode(), width, height, spp)
        if not h:
            raise RuntimeError("oracle: " + L.orc_last_error().decode())
        self.h = C.c_void_p(h)

    def info(self, si=0):
        out = (C.c_int * 8)()
        lib().orc_scene_info(self.h, si, out)
        k = ["shapes", "prims", "emitters", "width", "height", "spp", "nodes", "border"]
        return dict(zip(k, list(out)))

    def bvh_stats(self, si=0):
        sah = C.c_float()
        st = (C.c_int * 4)()
        lib().orc_bvh_stats(self.h, si, C.byref(sah), st)
        return dict(sah=sah.value, nodes=st[0], leaves=st[1], max_leaf=st[2], depth=st[3])

    def bvh(self, si=0):
        i = self.info(si)
        nodes = np.zeros((i["nodes"], 8), np.uint32)
        idx = np.zeros(i["prims"], np.uint32)
        lib().orc_bvh_copy(self.h, si, _p(nodes), _p(idx))
        return nodes, idx

    def shape(self, s, si=0):
        out = (C.c_int * 8)()
        lib().orc_shape_info(self.h, si, s, out)
        d = dict(zip(["type", "nV", "nF", "hasN", "hasUV", "emitter", "bsdf_type", "prim_offset"], list(out)))
        if d["type"] == 0:
            V = np.zeros((d["nV"], 3), np.float32)
            N = np.zeros((d["nV"], 3), np.float32) if d["hasN"] else None
            UV = np.zeros((d["nV"], 2), np.float32) if d["hasUV"] else None
            F = np.zeros((d["nF"], 3), np.uint32)
            cdf = np.zeros(d["nF"] + 1, np.float32)
            lib().orc_shape_copy(self.h, si, s, _p(V), _p(N), _p(UV), _p(F), _p(cdf))
            d.update(V=V, N=N, UV=UV, F=F, cdf=cdf)
        return d

    def images(self, si=0):
        """the scene's images as the device reads them: dicts of kind, lookup, cols, rows, scale, delta, texels (bytes)"""
        out = []
        for k in range(lib().orc_image_count(self.h, si)):
            info = (C.c_int * 4)()
            xf = np.zeros(4, np.float32)
            lib().orc_image_get(self.h, si, k, info, _p(xf), None)
            tex = np.zeros(info[2] * info[3] * 4, np.uint8)
            lib().orc_image_get(self.h, si, k, info, _p(xf), _p(tex))
            out.append(dict(kind=info[0], lookup=info[1], cols=info[2], rows=info[3], scale=xf[0:2].copy(), delta=xf[2:4].copy(), texels=tex))
        return out

    def shape_images(self, s, si=0):
        out = (C.c_int * 2)()
        dm = np.zeros(3, np.float32)
        lib().orc_shape_images(self.h, si, s, out, _p(dm))
        return dict(albedo_tex=out[0], normal_tex=out[1], dmfp=dm)

    def image_eval(self, k, uv, nrm=None, si=0):
        uv = np.ascontiguousarray(uv, np.float32)
        nrm = None if nrm is None else np.ascontiguousarray(nrm, np.float32)
        out = np.zeros((len(uv), 3), np.float32)
        lib().orc_image_eval(self.h, si, k, len(uv), _p(uv), _p(nrm), _p(out))
        return out

    def film_mode(self, splat):
        lib().orc_film_mode(self.h, int(bool(splat)))

    def film_filter_table(self):
        t = np.zeros(33, np.float32)
        r = C.c_float()
        lib().orc_film_filter_table(self.h, _p(t), C.byref(r))
        return t, r.value

    def camera(self, si=0):
        s2c = np.zeros((4, 4), np.float32)
        c2w = np.zeros((4, 4), np.float32)
        misc = np.zeros(4, np.float32)
        lib().orc_camera(self.h, si, _p(s2c), _p(c2w), _p(misc))
        return s2c, c2w, misc

    def camera_rays(self, pixel_samples, si=0):
        ps = np.ascontiguousarray(pixel_samples, np.float32)
        rays = np.zeros((len(ps), 8), np.float32)
        lib().orc_camera_rays(self.h, si, _p(ps), len(ps), _p(rays))
        return rays

    def trace(self, rays, any_hit=False, si=0, full=False, counters=False):
        rays = np.ascontiguousarray(rays, np.float32)
        hits = np.zeros((len(rays), 4), np.uint32)
        fu = np.zeros((len(rays), 16), np.float32) if full else None
        ctr = np.zeros(2, np.uint64) if counters else None
        lib().orc_trace(self.h, si, _p(rays), len(rays), int(any_hit), _p(hits), _p(fu), _p(ctr))
        r = [hits]
        if full:
            r.append(fu)
        if counters:
            r.append(ctr)
        return r[0] if len(r) == 1 else tuple(r)

    def li_samples(self, px, py, states, seqs, si=0):
        n = len(px)
        px = np.ascontiguousarray(px, np.int32)
        py = np.ascontiguousarray(py, np.int32)
        st = np.ascontiguousarray(states, np.uint64)
        sq = np.ascontiguousarray(seqs, np.uint64)
        rgb = np.zeros((n, 3), np.float32)
        ps = np.zeros((n, 2), np.float32)
        lib().orc_li_samples(self.h, si, n, _p(px), _p(py), _p(st), _p(sq), _p(rgb), _p(ps))
        return rgb, ps

    def li_rays(self, rays, states, seqs, si=0):
        rays = np.ascontiguousarray(rays, np.float32)
        st = np.ascontiguousarray(states, np.uint64)
        sq = np.ascontiguousarray(seqs, np.uint64)
        rgb = np.zeros((len(rays), 3), np.float32)
        lib().orc_li_rays(self.h, si, len(rays), _p(rays), _p(st), _p(sq), _p(rgb))
        return rgb

    def render(self, spp_begin, spp_end, mode=1, threads=8, clear=True):
        if clear:
            lib().orc_film_clear(self.h)
        st = np.zeros(5, np.uint64)
        sec = lib().orc_render(self.h, spp_begin, spp_end, mode, threads, _p(st))
        return sec, dict(zip(["closest", "shadow", "bounces", "nodes", "tris"], [int(x) for x in st]))

    def film(self):
        i = self.info()
        b = i["border"]
        f = np.zeros((i["height"] + 2 * b, i["width"] + 2 * b, 4), np.float32)
        lib().orc_film_copy(self.h, _p(f))
        return f

    def bitmap(self):
        i = self.info()
        img = np.zeros((i["height"], i["width"], 3), np.float32)
        lib().orc_bitmap(self.h, _p(img))
        return img

    # --- t-test suites
    def test_count(self):
        return lib().orc_test_count(self.h)

    def test_meta(self, i):
        L = lib()
        return dict(reference=L.orc_test_reference(self.h, i), path_mis=bool(L.orc_test_is_path_mis(self.h, i)),
                    integrator={0: "path_mis", 1: "path_mats", 2: "direct_mis"}.get(L.orc_test_integrator(self.h, i)),
                    n=L.orc_test_sample_count(self.h), significance=L.orc_test_significance(self.h))

    def test_run_one(self, i, state, seq, n):
        m = C.c_double()
        v = C.c_double()
        lib().orc_test_run_one(self.h, i, C.c_uint64(state), C.c_uint64(seq), n, C.byref(m), C.byref(v))
        return m.value, v.value


def ss_sample_dx(dmfp, p, nrm, states, seqs):
    """Subsurface::sampleDx alone: rows of probe o, d, mint, maxt, dx"""
    p = np.ascontiguousarray(p, np.float32)
    nrm = np.ascontiguousarray(nrm, np.float32)
    st = np.ascontiguousarray(states, np.uint64)
    sq = np.ascontiguousarray(seqs, np.uint64)
    out = np.zeros((len(p), 10), np.float32)
    lib().orc_ss_sample_dx(_p(np.ascontiguousarray(dmfp, np.float32)), len(p), _p(p), _p(nrm), _p(st), _p(sq), _p(out))
    return out


def pcg32(state, seq, n):
    out = np.zeros(n, np.uint32)
    lib().orc_pcg32(C.c_uint64(state), C.c_uint64(seq), n, _p(out))
    return out


def pcg32_floats(state, seq, n):
    out = np.zeros(n, np.float32)
    lib().orc_pcg32_floats(C.c_uint64(state), C.c_uint64(seq), n, _p(out))
    return out


def bsdf_sample(btype, params, wi, sample):
    wi = np.ascontiguousarray(wi, np.float32)
    sample = np.ascontiguousarray(sample, np.float32)
    params = np.ascontiguousarray(params, np.float32)
    out = np.zeros((len(wi), 12), np.float32)
    lib().orc_bsdf_sample(btype, _p(params), len(wi), _p(wi), _p(sample), _p(out))
    return out


def film_put(w, h, radius, pos, rgb):
    pos = np.ascontiguousarray(pos, np.float32)
    rgb = np.ascontiguousarray(rgb, np.float32)
    b = int(np.ceil(radius - 0.5))
    out = np.zeros((h + 2 * b, w + 2 * b, 4), np.float32)
    lib().orc_film_put(w, h, C.c_float(radius), len(pos), _p(pos), _p(rgb), _p(out))
    return out


def nl_means(img, r, f, sigma, h):
    img = np.ascontiguousarray(img, np.float32)
    out = np.zeros_like(img)
    lib().orc_nl_means(_p(img), _p(out), img.shape[1], img.shape[0], r, f, C.c_float(sigma), C.c_float(h))
    return out
