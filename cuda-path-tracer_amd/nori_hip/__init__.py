"""ctypes bindings of the MI355X path_mis renderer.

Two in-tree shared libraries are bound here, nothing else:
  libnori_hip.so   gfx950 kernels behind the device C ABI (include/nori_hip.h)
  libnori_host.so  Nori's host plugin surface (include/nori_host.h)
There is no CPU fallback: if the libraries are missing or no GPU is present, calls raise NoriHipError.
"""
import ctypes as C
import os
import numpy as np

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# A render call may run as two pipelines on two streams, which only run side by side on two hardware queues; the HIP runtime gives a
# process GPU_MAX_HW_QUEUES of them per device (4 unless set) and lets further streams share.  The library takes its second stream for
# the duration of a call only (csrc/nori_hip.hip, render_entry), so that idle contexts do not hold places; a process that keeps many
# streams busy at once wants more queues, and says so before the runtime starts (the variable is read once).  Unset: eight.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


class NoriHipError(RuntimeError):
    pass


class Camera(C.Structure):
    _fields_ = [("sample_to_camera", C.c_float * 16), ("camera_to_world", C.c_float * 16), ("width", C.c_int32), ("height", C.c_int32),
                ("near_clip", C.c_float), ("far_clip", C.c_float), ("filter_radius", C.c_float)]


class EnvMap(C.Structure):
    _fields_ = [("texture", C.c_uint32), ("brightness", C.c_float), ("frame", C.c_float * 9)]


class SceneDesc(C.Structure):
    _fields_ = [("n_nodes", C.c_uint32), ("nodes", C.c_void_p), ("n_prims", C.c_uint32), ("prims", C.c_void_p), ("shade_prims", C.c_void_p),
                ("root", C.c_uint32), ("root_min", C.c_float * 3), ("root_max", C.c_float * 3),
                ("n_shapes", C.c_uint32), ("shapes", C.c_void_p), ("n_emitters", C.c_uint32), ("emitters", C.c_void_p),
                ("n_cdf", C.c_uint32), ("cdf", C.c_void_p), ("n_light_prims", C.c_uint32), ("light_prims", C.c_void_p), ("camera", Camera),
                ("n_textures", C.c_uint32), ("textures", C.c_void_p), ("n_tex_bytes", C.c_uint64), ("tex_data", C.c_void_p), ("env", EnvMap)]


class Stats(C.Structure):
    _fields_ = [("camera_samples", C.c_uint64), ("closest_rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("invalid_samples", C.c_uint64),
                ("boxes_tested", C.c_uint64), ("prims_tested", C.c_uint64), ("iterations", C.c_uint64), ("render_ms", C.c_double),
                ("extend_ms", C.c_double), ("extend_launches", C.c_uint64), ("extend_rays", C.c_uint64), ("extend_boxes", C.c_uint64),
                ("extend_prims", C.c_uint64), ("extend_visits", C.c_uint64), ("extend_width", C.c_uint64), ("shadow_rays_culled", C.c_uint64),
                ("ss_paths_dropped", C.c_uint64), ("extend_node_bytes", C.c_uint64), ("pipelines", C.c_uint64)]


NODE_DTYPE = np.dtype([("lmin", "<f4", 3), ("lmax", "<f4", 3), ("rmin", "<f4", 3), ("rmax", "<f4", 3), ("left", "<u4"), ("right", "<u4"),
                       ("lcount", "<u4"), ("rcount", "<u4")])
PRIM_DTYPE = np.dtype([("a", "<f4", 3), ("e1", "<f4", 3), ("e2", "<f4", 3), ("prim", "<u4"), ("shape_flags", "<u4"), ("pad", "<u4")])
SHADE_DTYPE = np.dtype([("p0", "<f4", 3), ("p1", "<f4", 3), ("p2", "<f4", 3), ("n0", "<f4", 3), ("n1", "<f4", 3), ("n2", "<f4", 3),
                        ("uv0", "<f4", 2), ("uv1", "<f4", 2), ("uv2", "<f4", 2)])
SHAPE_DTYPE = np.dtype([("type", "<i4"), ("bsdf", "<i4"), ("emitter", "<i4"), ("flags", "<u4"), ("albedo", "<f4", 3), ("alpha", "<f4"),
                        ("int_ior", "<f4"), ("ext_ior", "<f4"), ("ks", "<f4"), ("center", "<f4", 3), ("radius", "<f4"), ("albedo_tex", "<u2"),
                        ("normal_tex", "<u2")])         # subsurface shapes: alpha holds ss_cs, ks holds ss_cd (unions in nori_hip.h)
TEXTURE_DTYPE = np.dtype([("kind", "<u4"), ("lookup", "<u4"), ("width", "<u4"), ("height", "<u4"), ("offset", "<u8"), ("scale", "<f4", 2), ("delta", "<f4", 2)])
EMITTER_DTYPE = np.dtype([("radiance", "<f4", 3), ("shape", "<i4"), ("cdf_offset", "<u4"), ("n_tris", "<u4"), ("light_prim_offset", "<u4"),
                          ("area_pdf", "<f4")])
NODE32_DTYPE = np.dtype([("w0", "<u4"), ("w1", "<u4"), ("bmin", "<f4", 3), ("bmax", "<f4", 3)])
RAY_DTYPE = np.dtype([("o", "<f4", 3), ("mint", "<f4"), ("d", "<f4", 3), ("maxt", "<f4")])
HIT_DTYPE = np.dtype([("prim", "<u4"), ("t", "<f4"), ("u", "<f4"), ("v", "<f4")])
assert NODE_DTYPE.itemsize == 64 and PRIM_DTYPE.itemsize == 48 and SHADE_DTYPE.itemsize == 96 and SHAPE_DTYPE.itemsize == 64 and EMITTER_DTYPE.itemsize == 32 and TEXTURE_DTYPE.itemsize == 40

MISS = 0xFFFFFFFF
_hip = None
_host = None


def _vp(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def hip_lib():
    """libnori_hip.so; raises if it is not built"""
    global _hip
    if _hip is None:
        p = os.path.join(_PKG, "libnori_hip.so")
        if not os.path.exists(p):
            raise NoriHipError("libnori_hip.so is not built: run __graft_entry__.build() or `make -C cuda-path-tracer_amd`")
        L = C.CDLL(p, mode=C.RTLD_GLOBAL)
        L.nori_hip_last_error.restype = C.c_char_p
        vp, sz, i32 = C.c_void_p, C.c_size_t, C.c_int
        L.nori_hip_create.argtypes = [i32, C.POINTER(vp)]
        L.nori_hip_destroy.argtypes = [vp]
        L.nori_hip_upload_scene.argtypes = [vp, vp]
        L.nori_hip_upload_scene_build.argtypes = [vp, vp]
        L.nori_hip_set_film_filter.argtypes = [vp, vp, C.c_float]
        L.nori_hip_set_integrator.argtypes = [vp, i32]
        L.nori_hip_build_info.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_double)]
        L.nori_hip_trace.argtypes = [vp, vp, sz, i32, vp, vp]
        L.nori_hip_trace_device.argtypes = [vp, vp, sz, i32, vp]
        L.nori_hip_camera_rays.argtypes = [vp, vp, sz, vp]
        L.nori_hip_li_samples.argtypes = [vp, sz, vp, vp, vp, vp, vp, vp]
        L.nori_hip_render.argtypes = [vp, i32, i32, i32, i32, i32, i32]
        L.nori_hip_render_tiles.argtypes = [vp, vp, i32, i32, i32]
        L.nori_hip_clear_film.argtypes = [vp]
        L.nori_hip_film_dims.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
        L.nori_hip_read_film.argtypes = [vp, vp]
        L.nori_hip_write_film.argtypes = [vp, vp]
        L.nori_hip_film_device_ptr.argtypes = [vp, C.POINTER(vp), C.POINTER(sz)]
        L.nori_hip_read_bitmap.argtypes = [vp, vp]
        L.nori_hip_filter.argtypes = [vp, i32, i32, C.c_float, C.c_float, vp]
        L.nori_hip_filter_noalloc.argtypes = [vp, i32, i32, C.c_float, C.c_float, vp]
        L.nori_hip_stats.argtypes = [vp, C.POINTER(Stats)]
        L.nori_hip_reset_stats.argtypes = [vp]
        L.nori_hip_stream.argtypes = [vp, C.POINTER(vp)]
        L.nori_hip_synchronize.argtypes = [vp]
        L.nori_hip_bsdf_vectors.argtypes = [vp, vp, sz, vp, vp, vp, vp, vp]
        L.nori_hip_pcg32.argtypes = [vp, C.c_uint64, C.c_uint64, sz, vp, vp]
        L.nori_hip_texture_eval.argtypes = [vp, C.c_uint32, sz, vp, vp, vp]
        L.nori_hip_device_count.argtypes = [C.POINTER(i32)]
        L.nori_hip_group_create.argtypes = [vp, i32, C.POINTER(vp)]
        L.nori_hip_group_destroy.argtypes = [vp]
        L.nori_hip_group_size.argtypes = [vp]
        L.nori_hip_group_context.restype = vp
        L.nori_hip_group_context.argtypes = [vp, i32]
        L.nori_hip_group_upload_scene.argtypes = [vp, vp, i32]
        L.nori_hip_group_set_integrator.argtypes = [vp, i32]
        L.nori_hip_group_set_film_filter.argtypes = [vp, vp, C.c_float]
        L.nori_hip_group_clear_film.argtypes = [vp]
        L.nori_hip_group_render.argtypes = [vp, i32, i32, i32]
        L.nori_hip_group_reduce_films.argtypes = [vp]
        L.nori_hip_group_reduce_ms.argtypes = [vp, C.POINTER(C.c_double)]
        L.nori_hip_group_plan.argtypes = [i32, i32, i32, i32, i32, i32, C.POINTER(i32), C.POINTER(i32), vp, i32, C.POINTER(i32)]
        _hip = L
    return _hip


def host_lib():
    global _host
    if _host is None:
        hip_lib()
        p = os.path.join(_PKG, "libnori_host.so")
        if not os.path.exists(p):
            raise NoriHipError("libnori_host.so is not built: run __graft_entry__.build() or `make -C cuda-path-tracer_amd`")
        L = C.CDLL(p)
        vp, i32 = C.c_void_p, C.c_int
        L.nori_host_last_error.restype = C.c_char_p
        L.nori_host_load.restype = vp
        L.nori_host_load.argtypes = [C.c_char_p, i32, i32, i32]
        L.nori_host_free.argtypes = [vp]
        L.nori_host_is_test.argtypes = [vp]
        L.nori_host_scene_count.argtypes = [vp]
        L.nori_host_scene_begin.restype = vp
        L.nori_host_scene_add_mesh.argtypes = [vp, C.c_char_p, vp, C.c_size_t, vp, C.c_size_t, vp, vp, C.c_char_p, vp, C.c_float, C.c_float, C.c_float, vp]
        L.nori_host_scene_set_camera.argtypes = [vp, vp, vp, vp, C.c_float, i32, i32]
        L.nori_host_scene_end.argtypes = [vp, i32, i32]
        L.nori_host_info.argtypes = [vp, i32, vp]
        L.nori_host_bvh_build_seconds.restype = C.c_double
        L.nori_host_bvh_build_seconds.argtypes = [vp, i32]
        L.nori_host_bvh32.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(C.c_uint32), C.POINTER(vp), C.POINTER(C.c_uint32)]
        L.nori_host_mesh.argtypes = [vp, i32, i32, vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
        L.nori_host_flatten.restype = C.POINTER(SceneDesc)
        L.nori_host_flatten.argtypes = [vp, i32]
        L.nori_host_preprocess.argtypes = [vp, i32, i32]
        L.nori_host_context.restype = vp
        L.nori_host_context.argtypes = [vp, i32]
        L.nori_host_render.argtypes = [vp, i32, i32, i32]
        L.nori_host_test_meta.argtypes = [vp, i32, C.POINTER(C.c_float), C.POINTER(i32), C.POINTER(C.c_float), C.POINTER(i32)]
        L.nori_host_test_run.argtypes = [vp, i32, vp]
        L.nori_host_bsdf_tests_run.argtypes = [vp, vp, i32]
        L.nori_host_li.argtypes = [vp, i32, C.c_size_t, vp, vp]
        L.nori_host_render_file.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
        L.nori_host_write_image.argtypes = [C.c_char_p, vp, i32, i32]
        _host = L
    return _host


def _hchk(rc):
    if rc != 0:
        raise NoriHipError(hip_lib().nori_hip_last_error().decode())


def _ochk(rc):
    if rc < 0:
        raise NoriHipError(host_lib().nori_host_last_error().decode())


def _from_ptr(ptr, n, dtype):
    if not ptr or n == 0:
        return np.zeros(0, dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=n)


class Context:
    """One NoriHipContext (one GPU).  Thin, allocation-free view over the device C ABI."""

    def __init__(self, handle=None, device=0, owned=True):
        self.L = hip_lib()
        self.owned = owned
        if handle is None:
            h = C.c_void_p()
            _hchk(self.L.nori_hip_create(device, C.byref(h)))
            handle = h.value
        self.h = C.c_void_p(handle)

    def close(self):
        if self.owned and self.h:
            self.L.nori_hip_destroy(self.h)
        self.h = None

    def upload(self, desc_ptr):
        _hchk(self.L.nori_hip_upload_scene(self.h, C.cast(desc_ptr, C.c_void_p)))

    def upload_build(self, desc_ptr):
        """upload with the tree built on the device (nori_hip_upload_scene_build)"""
        _hchk(self.L.nori_hip_upload_scene_build(self.h, C.cast(desc_ptr, C.c_void_p)))

    def set_film_filter(self, table33=None, radius=0.0):
        """filter-weighted splat (upstream Nori's ImageBlock::put) with the given 33-entry table, or None for the fork's film"""
        t = None if table33 is None else np.ascontiguousarray(table33, np.float32)
        _hchk(self.L.nori_hip_set_film_filter(self.h, _vp(t), float(radius)))

    def build_info(self):
        a, b, c, ms = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_double()
        _hchk(self.L.nori_hip_build_info(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(ms)))
        return dict(n_nodes=a.value, n_used=b.value, depth=c.value, build_ms=ms.value)

    def trace(self, rays, any_hit=False, full=False):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        hits = np.zeros(len(rays), HIT_DTYPE)
        fu = np.zeros((len(rays), 16), np.float32) if full else None
        _hchk(self.L.nori_hip_trace(self.h, _vp(rays), len(rays), int(any_hit), _vp(hits), _vp(fu)))
        return (hits, fu) if full else hits

    def camera_rays(self, pixel_samples):
        ps = np.ascontiguousarray(pixel_samples, np.float32).reshape(-1, 2)
        rays = np.zeros((len(ps), 8), np.float32)
        _hchk(self.L.nori_hip_camera_rays(self.h, _vp(ps), len(ps), _vp(rays)))
        return rays

    def li_samples(self, states, seqs, px=None, py=None, rays=None):
        st = np.ascontiguousarray(states, np.uint64)
        sq = np.ascontiguousarray(seqs, np.uint64)
        n = len(st)
        px = None if px is None else np.ascontiguousarray(px, np.int32)
        py = None if py is None else np.ascontiguousarray(py, np.int32)
        rays = None if rays is None else np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        rgb = np.zeros((n, 3), np.float32)
        _hchk(self.L.nori_hip_li_samples(self.h, n, _vp(px), _vp(py), _vp(st), _vp(sq), _vp(rays), _vp(rgb)))
        return rgb

    def bsdf_vectors(self, kind, params, wi, sample=None, wo=None, sample2=None):
        """nori_hip_bsdf_vectors; params = albedo/kd rgb, alpha, intIOR, extIOR (the layout of tests/golden/bsdf.npz)"""
        b = np.zeros(1, SHAPE_DTYPE)
        p = np.asarray(params, np.float32)
        b["bsdf"] = kind; b["albedo"] = p[0:3]; b["alpha"] = p[3]; b["int_ior"] = p[4]; b["ext_ior"] = p[5]; b["ks"] = np.float32(1.0) - p[0:3].max()
        b["emitter"] = -1
        wi = np.ascontiguousarray(wi, np.float32).reshape(-1, 3)
        sample = None if sample is None else np.ascontiguousarray(sample, np.float32).reshape(-1, 2)
        sample2 = None if sample2 is None else np.ascontiguousarray(sample2, np.float32).reshape(-1, 2)
        wo = None if wo is None else np.ascontiguousarray(wo, np.float32).reshape(-1, 3)
        out = np.zeros((len(wi), 12), np.float32)
        _hchk(self.L.nori_hip_bsdf_vectors(self.h, _vp(b), len(wi), _vp(wi), _vp(sample), _vp(sample2), _vp(wo), _vp(out)))
        return out

    def texture_eval(self, k, uv, nrm=None):
        """nori_hip_texture_eval: textures[k] (0-based) of the uploaded scene at uv; nrm for normal modifiers"""
        uv = np.ascontiguousarray(uv, np.float32).reshape(-1, 2)
        nrm = None if nrm is None else np.ascontiguousarray(nrm, np.float32).reshape(-1, 3)
        out = np.zeros((len(uv), 3), np.float32)
        _hchk(self.L.nori_hip_texture_eval(self.h, k, len(uv), _vp(uv), _vp(nrm), _vp(out)))
        return out

    def pcg32(self, state, seq, n, floats=False):
        out = np.zeros(n, np.float32 if floats else np.uint32)
        _hchk(self.L.nori_hip_pcg32(self.h, state, seq, n, None if floats else _vp(out), _vp(out) if floats else None))
        return out

    def render(self, spp_begin, spp_end, tile=None):
        if tile is None:
            rows, cols, b = self.film_dims()
            tile = (0, 0, cols - 2 * b, rows - 2 * b)
        _hchk(self.L.nori_hip_render(self.h, tile[0], tile[1], tile[2], tile[3], spp_begin, spp_end))

    def render_tiles(self, spp_begin, spp_end, tiles):
        """tiles: [(x0, y0, x1, y1), ...] rendered in one pass (nori_hip_render_tiles)"""
        t = np.ascontiguousarray(tiles, np.int32).reshape(-1, 4)
        if len(t) == 0:
            return                                     # a rank without tiles has nothing to render
        _hchk(self.L.nori_hip_render_tiles(self.h, _vp(t), len(t), spp_begin, spp_end))

    def trace_kernel_name(self):
        """the kernel bench.py's roofline is quoted on (the trace stage of the wavefront pipeline)"""
        if os.environ.get("NORI_HIP_MERGE", "1") == "0":
            return "k_trace_pool<TK_EXTEND>"
        st = self.stats()
        if st["extend_width"] == 4 and st["extend_node_bytes"] == 64:
            return "k_trace_pool4q"                                   # 64-byte quantised four-box nodes (k_trace_pool4q_leaf2)
        return {8: "k_trace_pool8", 4: "k_trace_pool4"}.get(st["extend_width"], "k_trace_pool<TK_BOTH>")

    def clear_film(self):
        _hchk(self.L.nori_hip_clear_film(self.h))

    def film_dims(self):
        r, c, b = C.c_int(), C.c_int(), C.c_int()
        _hchk(self.L.nori_hip_film_dims(self.h, C.byref(r), C.byref(c), C.byref(b)))
        return r.value, c.value, b.value

    def read_film(self):
        r, c, _ = self.film_dims()
        f = np.zeros((r, c, 4), np.float32)
        _hchk(self.L.nori_hip_read_film(self.h, _vp(f)))
        return f

    def write_film(self, film):
        film = np.ascontiguousarray(film, np.float32)
        _hchk(self.L.nori_hip_write_film(self.h, _vp(film)))

    def film_device_ptr(self):
        p, n = C.c_void_p(), C.c_size_t()
        _hchk(self.L.nori_hip_film_device_ptr(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def read_bitmap(self):
        r, c, b = self.film_dims()
        img = np.zeros((r - 2 * b, c - 2 * b, 3), np.float32)
        _hchk(self.L.nori_hip_read_bitmap(self.h, _vp(img)))
        return img

    def filter(self, r=3, f=10, sigma=0.5, h=0.002):
        rows, cols, b = self.film_dims()
        out = np.zeros((rows - 2 * b, cols - 2 * b, 3), np.float32)
        _hchk(self.L.nori_hip_filter(self.h, r, f, sigma, h, _vp(out)))
        return out

    def filter_noalloc(self, d_out, r=3, f=10, sigma=0.5, h=0.002):
        """nori_hip_filter_noalloc: d_out is a device address (height x width x 3 floats); asynchronous on the context's stream"""
        _hchk(self.L.nori_hip_filter_noalloc(self.h, r, f, sigma, h, C.c_void_p(d_out)))

    def stats(self):
        s = Stats()
        _hchk(self.L.nori_hip_stats(self.h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in Stats._fields_}

    def reset_stats(self):
        _hchk(self.L.nori_hip_reset_stats(self.h))

    def stream(self):
        p = C.c_void_p()
        _hchk(self.L.nori_hip_stream(self.h, C.byref(p)))
        return p.value

    def synchronize(self):
        _hchk(self.L.nori_hip_synchronize(self.h))


PARTITION_SPP, PARTITION_TILES = 0, 1


def group_plan(n_members, member, spp_begin, spp_end, height, partition):
    """nori_hip_group_plan: (spp_begin, spp_end, [(y0, y1), ...]) of one member of a group render; touches no device"""
    L = hip_lib()
    b, e, nb = C.c_int(), C.c_int(), C.c_int()
    cap = (max(height, 1) + 31) // 32
    bands = np.zeros((cap, 2), np.int32)
    _hchk(L.nori_hip_group_plan(n_members, member, spp_begin, spp_end, height, partition, C.byref(b), C.byref(e), _vp(bands), cap, C.byref(nb)))
    return b.value, e.value, [tuple(int(v) for v in bands[k]) for k in range(nb.value)]


def device_count():
    n = C.c_int()
    _hchk(hip_lib().nori_hip_device_count(C.byref(n)))
    return n.value


class Group:
    """nori_hip_group_*: one context per listed GPU of this node, driven together; films summed with RCCL"""

    def __init__(self, devices):
        self.L = hip_lib()
        d = np.ascontiguousarray(devices, np.int32)
        h = C.c_void_p()
        _hchk(self.L.nori_hip_group_create(_vp(d), len(d), C.byref(h)))
        self.h = h

    def close(self):
        if self.h:
            self.L.nori_hip_group_destroy(self.h)
        self.h = None

    def __len__(self):
        return self.L.nori_hip_group_size(self.h)

    def context(self, i=0):
        c = self.L.nori_hip_group_context(self.h, i)
        if not c:
            raise NoriHipError("no such member")
        return Context(handle=c, owned=False)

    def upload(self, desc_ptr, build_on_device=False):
        _hchk(self.L.nori_hip_group_upload_scene(self.h, C.cast(desc_ptr, C.c_void_p), int(build_on_device)))

    def clear_film(self):
        _hchk(self.L.nori_hip_group_clear_film(self.h))

    def render(self, spp_begin, spp_end, partition=PARTITION_SPP):
        _hchk(self.L.nori_hip_group_render(self.h, spp_begin, spp_end, partition))

    def reduce_films(self):
        _hchk(self.L.nori_hip_group_reduce_films(self.h))

    def reduce_ms(self):
        ms = C.c_double()
        _hchk(self.L.nori_hip_group_reduce_ms(self.h, C.byref(ms)))
        return ms.value


class HostScene:
    """A scene file (or <test> file) loaded through the host plugin surface."""

    def __init__(self, xml=None, width=0, height=0, spp=0, handle=None):
        self.L = host_lib()
        if handle is None:
            handle = self.L.nori_host_load(os.fspath(xml).encode(), width, height, spp)
            if not handle:
                raise NoriHipError(self.L.nori_host_last_error().decode())
        self.h = C.c_void_p(handle)

    def close(self):
        if self.h:
            self.L.nori_host_free(self.h)
        self.h = None

    @property
    def is_test(self):
        return bool(self.L.nori_host_is_test(self.h))

    @property
    def scene_count(self):
        return self.L.nori_host_scene_count(self.h)

    def info(self, scene=0):
        out = np.zeros(8, np.int32)
        _ochk(self.L.nori_host_info(self.h, scene, _vp(out)))
        return dict(zip(["shapes", "prims", "emitters", "width", "height", "spp", "nodes", "wide_nodes"], [int(x) for x in out]))

    def bvh_build_seconds(self, scene=0):
        return self.L.nori_host_bvh_build_seconds(self.h, scene)

    def bvh32(self, scene=0):
        n, ni = C.c_uint32(), C.c_uint32()
        pn, pi = C.c_void_p(), C.c_void_p()
        _ochk(self.L.nori_host_bvh32(self.h, scene, C.byref(pn), C.byref(n), C.byref(pi), C.byref(ni)))
        return _from_ptr(pn.value, n.value, NODE32_DTYPE), _from_ptr(pi.value, ni.value, np.uint32)

    def mesh(self, shape, scene=0):
        counts = np.zeros(4, np.int32)
        V, N, UV, F, cdf = (C.c_void_p() for _ in range(5))
        rc = self.L.nori_host_mesh(self.h, scene, shape, _vp(counts), C.byref(V), C.byref(N), C.byref(UV), C.byref(F), C.byref(cdf))
        _ochk(rc)
        if rc == 1:
            return None
        nV, nF, hasN, hasUV = [int(x) for x in counts]
        return dict(V=_from_ptr(V.value, nV * 3, np.float32).reshape(-1, 3), N=_from_ptr(N.value, nV * 3, np.float32).reshape(-1, 3) if hasN else None,
                    UV=_from_ptr(UV.value, nV * 2, np.float32).reshape(-1, 2) if hasUV else None,
                    F=_from_ptr(F.value, nF * 3, np.uint32).reshape(-1, 3), cdf=_from_ptr(cdf.value, nF + 1, np.float32))

    def flatten(self, scene=0):
        d = self.L.nori_host_flatten(self.h, scene)
        if not d:
            raise NoriHipError(self.L.nori_host_last_error().decode())
        return d

    def tables(self, scene=0):
        """numpy views of the flattened tables"""
        d = self.flatten(scene).contents
        return dict(desc=d, nodes=_from_ptr(d.nodes, d.n_nodes, NODE_DTYPE), prims=_from_ptr(d.prims, d.n_prims, PRIM_DTYPE),
                    shade=_from_ptr(d.shade_prims, d.n_prims, SHADE_DTYPE), shapes=_from_ptr(d.shapes, d.n_shapes, SHAPE_DTYPE),
                    emitters=_from_ptr(d.emitters, d.n_emitters, EMITTER_DTYPE), cdf=_from_ptr(d.cdf, d.n_cdf, np.float32),
                    light_prims=_from_ptr(d.light_prims, d.n_light_prims, SHADE_DTYPE),
                    textures=_from_ptr(d.textures, d.n_textures, TEXTURE_DTYPE), tex_data=_from_ptr(d.tex_data, d.n_tex_bytes, np.uint8))

    def preprocess(self, scene=0, device=-1):
        _ochk(self.L.nori_host_preprocess(self.h, scene, device))
        return self.context(scene)

    def context(self, scene=0):
        c = self.L.nori_host_context(self.h, scene)
        if not c:
            raise NoriHipError("no device context: preprocess() has not run (" + self.L.nori_host_last_error().decode() + ")")
        return Context(handle=c, owned=False)

    def render(self, spp_begin, spp_end, scene=0):
        _ochk(self.L.nori_host_render(self.h, scene, spp_begin, spp_end))

    def test_meta(self, scene):
        ref, n, sig, pm = C.c_float(), C.c_int(), C.c_float(), C.c_int()
        _ochk(self.L.nori_host_test_meta(self.h, scene, C.byref(ref), C.byref(n), C.byref(sig), C.byref(pm)))
        return dict(reference=ref.value, n=n.value, significance=sig.value, path_mis=bool(pm.value))

    def li(self, rays, scene=0):
        """Integrator::Li, one virtual call per ray (the reference's per-ray interface)"""
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        rgb = np.zeros((len(rays), 3), np.float32)
        _ochk(self.L.nori_host_li(self.h, scene, len(rays), _vp(rays), _vp(rgb)))
        return rgb

    def bsdf_tests_run(self, max_tests=256):
        """chi2test / BSDF-mode ttest files: one row of 6 values per test (include/nori_host.h)"""
        out = np.zeros((max_tests, 6), np.float64)
        n = self.L.nori_host_bsdf_tests_run(self.h, _vp(out), max_tests)
        _ochk(n)
        return out[:n]

    def test_run(self, scene):
        out = np.zeros(7, np.float64)
        _ochk(self.L.nori_host_test_run(self.h, scene, _vp(out)))
        return dict(zip(["mean", "variance", "reference", "t", "pvalue", "alpha", "passed"], out.tolist()))


class SceneBuilder:
    """Programmatic scene (procedural meshes) through the same plugin classes the parser instantiates."""

    def __init__(self):
        self.L = host_lib()
        h = self.L.nori_host_scene_begin()
        if not h:
            raise NoriHipError(self.L.nori_host_last_error().decode())
        self.h = C.c_void_p(h)

    def add_mesh(self, name, V, F, N=None, UV=None, bsdf="diffuse", color=(0.5, 0.5, 0.5), alpha=0.1, int_ior=1.5046, ext_ior=1.000277, radiance=None):
        V = np.ascontiguousarray(V, np.float32).reshape(-1, 3)
        F = np.ascontiguousarray(F, np.uint32).reshape(-1, 3)
        N = None if N is None else np.ascontiguousarray(N, np.float32)
        UV = None if UV is None else np.ascontiguousarray(UV, np.float32)
        col = np.asarray(color, np.float32)
        rad = None if radiance is None else np.asarray(radiance, np.float32)
        _ochk(self.L.nori_host_scene_add_mesh(self.h, name.encode(), _vp(V), len(V), _vp(F), len(F), _vp(N), _vp(UV), bsdf.encode(), _vp(col), alpha, int_ior,
                                              ext_ior, _vp(rad)))

    def set_camera(self, origin, target, up, fov, width, height):
        o, t, u = (np.asarray(x, np.float32) for x in (origin, target, up))
        _ochk(self.L.nori_host_scene_set_camera(self.h, _vp(o), _vp(t), _vp(u), fov, width, height))

    def finish(self, spp=1, build_threads=0):
        _ochk(self.L.nori_host_scene_end(self.h, spp, build_threads))
        return HostScene(handle=self.h.value)


def write_image(path, rgb):
    """linear rgb (H x W x 3) to "*.png" (8-bit sRGB) or PFM"""
    rgb = np.ascontiguousarray(rgb, np.float32)
    _ochk(host_lib().nori_host_write_image(os.fspath(path).encode(), _vp(rgb), rgb.shape[1], rgb.shape[0]))


def render_file(xml):
    buf = C.create_string_buffer(4096)
    _ochk(host_lib().nori_host_render_file(os.fspath(xml).encode(), buf, 4096))
    return buf.value.decode()
