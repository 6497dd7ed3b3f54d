"""GPU: the allocation-free filter entry point (nori_hip_filter_noalloc) against nori_hip_filter, through the C ABI.

Two calls in a row on one GPU, the second on a smaller image (another context: every context of a GPU shares the
static scratch array, so the second call finds the first one's pixels in it).  Same kernels, same arithmetic: the
outputs are compared bit for bit (the bar for this floating-point path elsewhere is rtol 2e-4; equality is stricter)."""
import ctypes as C
import os

import numpy as np
import pytest

import nori_hip as NH
from conftest import SCENES

pytestmark = pytest.mark.gpu


class DeviceBuffer:
    """a caller-owned device buffer (the test may allocate; the entry point under test may not)"""

    def __init__(self, nbytes):
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        self.hip.hipFree.argtypes = [C.c_void_p]
        self.nbytes = nbytes
        p = C.c_void_p()
        assert self.hip.hipMalloc(C.byref(p), nbytes) == 0
        self.ptr = p.value
        assert self.hip.hipMemset(self.ptr, 0xff, nbytes) == 0          # NaN pattern: every float must be overwritten

    def to_host(self, shape):
        out = np.zeros(shape, np.float32)
        assert out.nbytes == self.nbytes
        assert self.hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), self.ptr, self.nbytes, 2) == 0      # hipMemcpyDeviceToHost
        return out

    def free(self):
        self.hip.hipFree(self.ptr)


def device_free_bytes():
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    free, total = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


def test_filter_noalloc_matches_filter_on_two_calls_in_a_row(monkeypatch):
    monkeypatch.delenv("NORI_HIP_FILTER", raising=False)               # (nori_hip_filter's other kernel is not what the entry point shares)
    xml = os.path.join(SCENES, "cbox", "cbox_path_mis.xml")
    for W, H, r, f in [(96, 80, 3, 10), (48, 40, 2, 4)]:
        hs = NH.HostScene(xml, W, H, 8)
        ctx = hs.preprocess()
        ctx.render(0, 8)
        ref = ctx.filter(r, f, 0.5, 0.002)                              # the existing entry point (host output)
        buf = DeviceBuffer(W * H * 3 * 4)
        ctx.synchronize()
        free_before = device_free_bytes()
        ctx.filter_noalloc(buf.ptr, r, f, 0.5, 0.002)
        ctx.synchronize()
        assert device_free_bytes() == free_before                       # the call took no device memory (and gave none back)
        out = buf.to_host((H, W, 3))
        buf.free()
        assert np.isfinite(out).all()
        assert np.array_equal(out, ref), np.abs(out - ref).max()
        hs.close()


def test_filter_noalloc_rejects_images_beyond_its_maximum():
    xml = os.path.join(SCENES, "cbox", "cbox_path_mis.xml")
    hs = NH.HostScene(xml, 1921, 1080, 1)                              # one column more than NORI_HIP_NOALLOC_MAX_PIXELS allows
    ctx = hs.preprocess()
    buf = DeviceBuffer(16)
    with pytest.raises(NH.NoriHipError, match="NORI_HIP_NOALLOC_MAX_PIXELS"):
        ctx.filter_noalloc(buf.ptr)
    buf.free()
    hs.close()
