"""ctypes bindings for oracle/_ref/libnori_ref.so (the reference's own sources) — build container only."""
import ctypes as C
import os
import re
import tempfile
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_ROOT = "/root/reference"
_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(os.path.join(HERE, "_ref", "libnori_ref.so"))
        L.ref_last_error.restype = C.c_char_p
        L.ref_load.restype = C.c_void_p
        L.ref_load.argtypes = [C.c_char_p, C.c_char_p]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def override_xml(xml_path, width=None, height=None, spp=None, integrator=None):
    """Writes a copy of a scene XML with camera size / sample count replaced (the reference reads them from XML)."""
    s = open(xml_path).read()
    if width is not None:
        s = re.sub(r'(<integer\s+name="width"\s+value=")\d+(")', r"\g<1>%d\g<2>" % width, s)
    if height is not None:
        s = re.sub(r'(<integer\s+name="height"\s+value=")\d+(")', r"\g<1>%d\g<2>" % height, s)
    if spp is not None:
        s = re.sub(r'(<integer\s+name="sampleCount"\s+value=")\d+(")', r"\g<1>%d\g<2>" % spp, s)
    if integrator is not None:
        s = re.sub(r'(<integrator\s+type=")\w+(")', r"\g<1>%s\g<2>" % integrator, s)
    f = tempfile.NamedTemporaryFile("w", suffix=".xml", delete=False)
    f.write(s)
    f.close()
    return f.name


class RefScene:
    def __init__(self, xml, resolve_dir=None):
        L = lib()
        h = L.ref_load(xml.encode(), (resolve_dir or os.path.dirname(xml)).encode())
        if not h:
            raise RuntimeError("ref: " + L.ref_last_error().decode())
        self.h = C.c_void_p(h)

    def info(self):
        out = (C.c_int * 6)()
        lib().ref_scene_info(self.h, out)
        return dict(zip(["shapes", "prims", "emitters", "width", "height", "spp"], list(out)))

    def mesh(self, s):
        out = (C.c_int * 4)()
        lib().ref_mesh_info(self.h, s, out)
        nV, nF, hasN, hasUV = list(out)
        V = np.zeros((nV, 3), np.float32)
        N = np.zeros((nV, 3), np.float32) if hasN else None
        UV = np.zeros((nV, 2), np.float32) if hasUV else None
        F = np.zeros((nF, 3), np.uint32)
        lib().ref_mesh_copy(self.h, s, _p(V), _p(N), _p(UV), _p(F))
        return dict(V=V, N=N, UV=UV, F=F)

    def camera_rays(self, ps):
        ps = np.ascontiguousarray(ps, np.float32)
        rays = np.zeros((len(ps), 8), np.float32)
        lib().ref_camera_rays(self.h, _p(ps), len(ps), _p(rays))
        return rays

    def trace(self, rays, any_hit=False, full=False):
        rays = np.ascontiguousarray(rays, np.float32)
        hits = np.zeros((len(rays), 4), np.uint32)
        fu = np.zeros((len(rays), 16), np.float32) if full else None
        lib().ref_trace(self.h, _p(rays), len(rays), int(any_hit), _p(hits), _p(fu))
        return (hits, fu) if full else hits

    def li_samples(self, px, py, states, seqs):
        n = len(px)
        a = [np.ascontiguousarray(x, np.int32) for x in (px, py, states, seqs)]
        rgb = np.zeros((n, 3), np.float32)
        ps = np.zeros((n, 2), np.float32)
        lib().ref_li_samples(self.h, n, _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), _p(rgb), _p(ps))
        return rgb, ps

    def li_rays(self, rays, states, seqs):
        rays = np.ascontiguousarray(rays, np.float32)
        st = np.ascontiguousarray(states, np.int32)
        sq = np.ascontiguousarray(seqs, np.int32)
        rgb = np.zeros((len(rays), 3), np.float32)
        lib().ref_li_rays(self.h, len(rays), _p(rays), _p(st), _p(sq), _p(rgb))
        return rgb

    def ttest(self, n, state, seq):
        m = C.c_double()
        v = C.c_double()
        lib().ref_ttest_scene(self.h, n, state, seq, C.byref(m), C.byref(v))
        return m.value, v.value


def bsdf_sample(btype, params, wi, sample):
    wi = np.ascontiguousarray(wi, np.float32)
    sample = np.ascontiguousarray(sample, np.float32)
    params = np.ascontiguousarray(params, np.float32)
    out = np.zeros((len(wi), 12), np.float32)
    lib().ref_bsdf_sample(btype, _p(params), len(wi), _p(wi), _p(sample), _p(out))
    return out


def ss_sample_dx(dmfp, p, nrm, states, seqs):
    """Subsurface::sampleDx alone: rows of probe o, d, mint, maxt, dx"""
    p = np.ascontiguousarray(p, np.float32)
    nrm = np.ascontiguousarray(nrm, np.float32)
    st = np.ascontiguousarray(states, np.uint64)
    sq = np.ascontiguousarray(seqs, np.uint64)
    out = np.zeros((len(p), 10), np.float32)
    lib().ref_ss_sample_dx(_p(np.ascontiguousarray(dmfp, np.float32)), len(p), _p(p), _p(nrm), _p(st), _p(sq), _p(out))
    return out


def pcg32(state, seq, n):
    out = np.zeros(n, np.uint32)
    lib().ref_pcg32(C.c_uint64(state), C.c_uint64(seq), n, _p(out))
    return out


def sampler_check():
    return lib().ref_sampler_check()
