"""GPU: the device's sampler and BSDF code alone, against the reference's known answers and vectors, and the reference's BSDF
test plugins (chi2test, ttest in BSDF mode) run on it."""
import os

import numpy as np
import pytest

import nori_hip as NH
import oracle_lib as O
from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = NH.Context(device=0)
    yield c
    c.close()


def test_device_pcg32_known_answers(ctx):
    # ext/pcg32/pcg32-demo.out:8 — seed(42, 54), the first six 32-bit outputs
    kat = [0xa15c02b7, 0x7b47f409, 0xba1d3330, 0x83d2f293, 0xbfa4784b, 0xcbed606e]
    assert ctx.pcg32(42, 54, 6).tolist() == kat
    # nextFloat, and a long stream against the oracle's generator (itself pinned to the same known answers)
    assert np.array_equal(ctx.pcg32(1234, 1, 100000, floats=True), O.pcg32_floats(1234, 1, 100000))
    assert np.array_equal(ctx.pcg32(2 ** 63 + 5, 2 ** 40 + 3, 4096), O.pcg32(2 ** 63 + 5, 2 ** 40 + 3, 4096))


def test_device_bsdfs_against_reference_vectors(ctx):
    """tests/golden/bsdf.npz: BSDF::sample / pdf / eval of the reference's own classes on 2048 (wi, sample) pairs.  Delta BSDFs,
    validity and delta flags, and every direction, weight, pdf and value of the diffuse and microfacet BSDFs are bit-exact: the device
    evaluates sin / cos / logf / expf / acos with glibc's algorithms."""
    g = np.load(os.path.join(GOLDEN, "bsdf.npz"))
    for k in g.files:
        if k.endswith("_type") or k.endswith("_params") or k in ("wi", "sample") or k.startswith("ss_"):
            continue
        out = ctx.bsdf_vectors(int(g[k + "_type"]), g[k + "_params"], g["wi"], g["sample"])
        ref = g[k]
        assert np.array_equal(out[:, 10:12], ref[:, 10:12]), k                              # isDelta, wo_valid
        ok = ref[:, 11] == 1
        if k.startswith("dielectric") or k == "mirror":
            # reflection is exact; refraction goes through a double sqrt and one normalisation: the same operations on both sides
            assert np.array_equal(out[ok, 0:6], ref[ok, 0:6], equal_nan=True), k
        elif k in ("diffuse", "subsurface"):
            # cosine-hemisphere sampling goes through sin and cos only, which the device evaluates with glibc's algorithm (nd_sincosf,
            # csrc/nori_device.h): direction, weight, pdf and value are the reference's bit for bit (round 3; 2e-6 before)
            assert np.array_equal(out[ok, 3:6], ref[ok, 3:6], equal_nan=True), k
            if k == "diffuse":
                assert np.array_equal(out[ok], ref[ok], equal_nan=True), k
        else:
            # microfacet: Beckmann sampling and density go through logf / expf as well, glibc's on the device too (nd_logf, nd_expf):
            # bit for bit (round 3; 2e-4 relative before)
            assert np.array_equal(out[ok], ref[ok], equal_nan=True), (k, np.abs(out[ok] - ref[ok]).max(0))
        # eval / pdf at given directions: the same entry with wo handed in reproduces what it reported after sampling
        again = ctx.bsdf_vectors(int(g[k + "_type"]), g[k + "_params"], g["wi"][ok], wo=out[ok, 3:6])
        assert np.array_equal(again[:, 6:10], out[ok, 6:10], equal_nan=True), k


CHI2_XML = """<?xml version="1.0" encoding="utf-8"?>
<test type="chi2test">
	<integer name="testCount" value="3"/>
	<bsdf type="diffuse"><color name="albedo" value="0.5 0.5 0.5"/></bsdf>
	<bsdf type="microfacet"><float name="alpha" value="0.3"/><color name="kd" value="0.2 0.25 0.1"/></bsdf>
	<bsdf type="microfacet"><float name="alpha" value="0.6"/><color name="kd" value="0.5 0.5 0.5"/></bsdf>
	<bsdf type="microfacet"/>
</test>
"""


def test_chi2test_plugin_on_the_device(tmp_path):
    """<test type="chi2test"> (src/chi2test.cpp): sample() against pdf() for diffuse and three Beckmann microfacet BSDFs (the last
    one with the reference's defaults, alpha 0.1), three incident directions each, 1 M samples per test, 10 x 20 cells,
    alpha = 0.01 Sidak-corrected over the 12 tests"""
    p = tmp_path / "chi2.xml"
    p.write_text(CHI2_XML)
    h = NH.HostScene(str(p))
    res = h.bsdf_tests_run()
    assert len(res) == 12
    assert (res[:, 1] > 5).all(), res                  # degrees of freedom
    assert (res[:, 4] == 1).all(), res                 # every test accepted
    h.close()


def test_ttest_plugin_in_bsdf_mode_on_the_device(tmp_path):
    """<test type="ttest"> with <bsdf> children (src/ttest.cpp:107-147).  References: the diffuse weight is its albedo (exact); the
    microfacet means are the oracle's over an independent stream (the oracle's BSDFs are bit-identical to the reference's)."""
    rng = np.random.default_rng(3)
    n = 400000
    refs = []
    params = np.array([0.2, 0.25, 0.1, 0.3, 1.5046, 1.000277], np.float32)
    for ang in (0.0, 30.0, 60.0):
        th = np.deg2rad(ang)
        wi = np.tile(np.array([np.sin(th), 0, np.cos(th)], np.float32), (n, 1))
        out = O.bsdf_sample(1, params, wi, rng.random((n, 2)).astype(np.float32))
        refs.append(float((out[:, 0] * 0.212671 + out[:, 1] * 0.715160 + out[:, 2] * 0.072169).mean()))
    for body, angles, references in [
            ('<bsdf type="diffuse"><color name="albedo" value="0.5 0.5 0.5"/></bsdf>', "0, 30, 60", "0.5, 0.5, 0.5"),
            ('<bsdf type="microfacet"><float name="alpha" value="0.3"/><color name="kd" value="0.2 0.25 0.1"/></bsdf>', "0, 30, 60", ", ".join("%.6f" % r for r in refs))]:
        p = tmp_path / "ttest.xml"
        p.write_text('<?xml version="1.0" encoding="utf-8"?>\n<test type="ttest">\n<string name="angles" value="%s"/>\n<string name="references" value="%s"/>\n%s\n</test>\n' % (angles, references, body))
        h = NH.HostScene(str(p))
        res = h.bsdf_tests_run()
        assert len(res) == 3 and (res[:, 4] == 1).all(), res
        h.close()
    # and a wrong reference is rejected
    p.write_text('<?xml version="1.0" encoding="utf-8"?>\n<test type="ttest">\n<string name="angles" value="30"/>\n<string name="references" value="%.6f"/>\n%s\n</test>\n' % (refs[1] * 1.02, body))
    h = NH.HostScene(str(p))
    assert h.bsdf_tests_run()[0, 4] == 0
    h.close()
