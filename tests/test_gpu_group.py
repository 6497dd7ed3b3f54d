"""GPU: several devices behind one handle (nori_hip_group_*, the `devices` property of the path_mis plugin).  The box that runs
these tests has one GPU, so the group has one member: the code path - contexts made by the group, per-device host threads'
work split, RCCL communicator, in-place all-reduce on the context's stream, accumulation across calls - is the one eight GPUs take."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import nori_hip as NH
from conftest import SCENES

pytestmark = pytest.mark.gpu
XML = os.path.join(SCENES, "cbox", "cbox_path_mis.xml")


def test_group_of_one_device_renders_reduces_and_accumulates_like_a_context():
    W, H, spp = 64, 72, 6
    hs = NH.HostScene(XML, W, H, spp)
    ctx = hs.preprocess()
    ctx.clear_film(); ctx.render(0, spp)
    want = ctx.read_film()
    g = NH.Group([0])
    assert len(g) == 1 and NH.device_count() >= 1
    g.upload(hs.flatten())
    for part in (NH.PARTITION_SPP, NH.PARTITION_TILES):
        g.clear_film()
        g.render(0, 2, part); g.render(2, spp, part)              # two calls accumulate, the earlier sum is counted once
        got = g.context(0).read_film()
        assert np.array_equal(got[..., 3], want[..., 3])
        assert np.allclose(got, want, rtol=1e-4, atol=1e-5)
        assert g.reduce_ms() >= 0.0
    # configs[4]'s order of operations: render slices, reduce, then the filter pass on member 0
    a, b = g.context(0).filter(2, 3, 0.5, 0.05), ctx.filter(2, 3, 0.5, 0.05)
    assert np.allclose(a, b, rtol=1e-3, atol=1e-5)
    with pytest.raises(NH.NoriHipError, match="listed twice"):
        NH.Group([0, 0])
    with pytest.raises(NH.NoriHipError):
        NH.Group([NH.device_count()])
    g.close(); hs.close()


def test_devices_property_of_the_plugin_takes_the_group_path():
    W, H, spp = 48, 40, 4
    hs = NH.HostScene(XML, W, H, spp)
    ctx = hs.preprocess(); ctx.render(0, spp)
    want = ctx.read_film()
    for part in ("spp", "tiles"):
        os.environ.update(NORI_HIP_DEVICES="all", NORI_HIP_PARTITION=part)
        try:
            hg = NH.HostScene(XML, W, H, spp)
            cg = hg.preprocess()
        finally:
            del os.environ["NORI_HIP_DEVICES"], os.environ["NORI_HIP_PARTITION"]
        hg.render(0, 1); hg.render(1, spp)                        # Integrator::render through nori_hip_group_render
        got = cg.read_film()
        assert np.array_equal(got[..., 3], want[..., 3]) and np.allclose(got, want, rtol=1e-4, atol=1e-5)
        hg.close()
    hs.close()


def test_per_ray_li_of_the_plugin_is_the_batched_li():
    """Integrator::Li(scene, sampler, ray), the reference's per-ray interface (include/nori/integrator.h:48-58), on the GPU plugin:
    call k owns the stream pcg32.seed(k, 0x4c69), so it equals nori_hip_li_samples on the same rays and streams"""
    hs = NH.HostScene(XML, 32, 32, 1)
    ctx = hs.preprocess()
    rng = np.random.default_rng(2)
    rays = ctx.camera_rays((rng.random((200, 2)) * 32).astype(np.float32))
    a = hs.li(rays)
    b = ctx.li_samples(np.arange(200, dtype=np.uint64), np.full(200, 0x4c69, np.uint64), rays=rays)
    assert np.array_equal(a, b) and a.sum() > 0
    hs.close()


def test_bench_group_driver_and_strong_scaling_prediction():
    """bench.py --driver group: one process, nori_hip_group_render over the devices present (one here, eight asked for), the same
    JSON line as the torchrun mode with the all-reduce's device time in phases_ms_per_step; and --as-rank N, the strong-scaling
    prediction that renders every rank's share of an N-way split on this GPU"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    def run(*args):
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *args], capture_output=True, text=True, timeout=300, cwd=root)
        assert out.returncode == 0, out.stderr[-2000:]
        return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    for part in ("spp", "tiles"):
        d = run("--driver", "group", "--gpus", "8", "--workload", "cbox", "--spp", "4", "--steps", "2", "--warmup", "1", "--no-cpu", "--partition", part)
        assert d["n_gpus"] == NH.device_count() >= 1 and d["unit"] == "Msamples/s" and d["value"] > 0
        assert d["scaling"] == ("strong" if part == "tiles" else "weak")
        assert set(d["phases_ms_per_step"]) == {"render", "reduce", "filter"} and d["phases_ms_per_step"]["reduce"] >= 0
        assert d["roofline"]["kernel_ms"] > 0 and "group" in d["config"]["driver"]
    d = run("--as-rank", "4", "--workload", "cbox", "--spp", "8", "--steps", "2", "--warmup", "1")
    assert d["n_ranks"] == 4 and d["ms_whole"] > 0
    for part in ("tiles", "spp"):
        p = d["partitions"][part]
        assert len(p["ms_per_rank"]) == 4 and 0 < p["predicted_efficiency"] <= 1.5
