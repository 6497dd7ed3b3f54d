"""GPU: BASELINE configs[4] - "4K progressive 16k-spp cbox split by spp over 8 x MI355X with RCCL radiance reduce + filter pass" -
at its image size, through the product's own multi-GPU path (nori_hip_group_*; the box has one GPU, so the group has one member:
same contexts, host threads, communicator, in-place all-reduce and order of operations as eight).

  * 3840 x 2160: two group renders that accumulate (spp partition, then tiles), the reduce of the 133 MB film, the NL-means pass
    with the CUDA fork's defaults (r 3, f 10, sigma 0.5, h 0.002: cuda/nori/scene.cpp:28-32 with its swapped keys) on member 0 -
    size-independent properties, and equality with a plain context's result
  * the same sequence at 96 x 96 against the oracle: film weights, image, and O.nl_means (the restatement of
    cuda/filter.cc.cu:53-110) on the oracle's own film
Reference: cuda/filter.cc.cu:125-176 after the film merge of src/block.cpp:131-140."""
import os

import numpy as np
import pytest

import nori_hip as NH
import oracle_lib as O
from conftest import SCENES

pytestmark = pytest.mark.gpu
XML = os.path.join(SCENES, "cbox", "cbox_path_mis.xml")
FORK = dict(r=3, f=10, sigma=0.5, h=0.002)


def to_srgb(v):
    v = np.asarray(v, np.float32)
    return np.where(v <= 0.0031308, 12.92 * v, 1.055 * np.power(np.maximum(v, 1e-12), 1 / 2.4) - 0.055).astype(np.float32)   # src/common.cpp:192-206


def test_c5_at_3840x2160_group_render_reduce_and_filter():
    W, H, spp = 3840, 2160, 2
    hs = NH.HostScene(XML, W, H, spp)
    g = NH.Group([0])
    g.upload(hs.flatten())
    g.clear_film()
    g.render(0, 1, NH.PARTITION_SPP)                       # samples [0, 1) by spp slices ...
    assert g.reduce_ms() >= 0.0
    g.render(1, 2, NH.PARTITION_TILES)                     # ... and [1, 2) by 32-row bands, accumulated onto the reduced film
    assert g.reduce_ms() >= 0.0
    c0 = g.context(0)
    rows, cols, border = c0.film_dims()
    assert (rows, cols) == (H + 2 * border, W + 2 * border)
    ptr, nbytes = c0.film_device_ptr()
    assert nbytes == rows * cols * 16 >= 132 * 10 ** 6       # the 133 MB film the all-reduce moves
    film = c0.read_film()
    assert np.isfinite(film).all()
    assert film[..., 3].sum(dtype=np.float64) == W * H * spp                  # every sample landed once (weights are integers)
    st = c0.stats()
    assert st["invalid_samples"] == 0
    out = c0.filter(**FORK)
    assert out.shape == (H, W, 3) and np.isfinite(out).all()
    # the same job on a plain context
    ctx = hs.preprocess()
    ctx.clear_film(); ctx.render(0, spp)
    want = ctx.read_film()
    assert np.array_equal(film[..., 3], want[..., 3])
    assert np.allclose(film, want, rtol=1e-4, atol=1e-5)
    ref = ctx.filter(**FORK)
    assert np.allclose(out, ref, rtol=1e-3, atol=1e-5)
    # what the filter did, whatever the size: a weighted mean of input pixels per output pixel, which keeps the image mean
    noisy = to_srgb(ctx.read_bitmap())
    assert out.min() >= noisy.min() - 1e-5 and out.max() <= noisy.max() * (1 + 1e-5)     # every output pixel is a convex combination of input pixels
    assert abs(out.mean() - noisy.mean()) < 0.02 * noisy.mean()
    g.close(); hs.close()


def test_c5_sequence_small_against_the_oracle():
    W, H, spp = 96, 96, 4
    hs = NH.HostScene(XML, W, H, spp)
    g = NH.Group([0])
    g.upload(hs.flatten())
    g.clear_film()
    g.render(0, 2, NH.PARTITION_SPP); g.render(2, spp, NH.PARTITION_TILES)
    c0 = g.context(0)
    film = c0.read_film()
    o = O.OracleScene(XML, W, H, spp)
    o.render(0, spp, mode=1, threads=8)
    ofilm, oimg = o.film(), o.bitmap()
    assert np.array_equal(film[..., 3], ofilm[..., 3])
    img = c0.read_bitmap()
    rel = np.abs(img.mean((0, 1)) - oimg.mean((0, 1))) / oimg.mean((0, 1))
    assert rel.max() < 2e-3, rel
    out = c0.filter(**FORK)
    # the filter alone, on identical input: the device's bitmap through the oracle's NL-means
    assert np.allclose(out, O.nl_means(to_srgb(img), **FORK), rtol=2e-4, atol=2e-5)
    # and end to end against the oracle's own film (the inputs differ by the device-libm ulps of a few samples, which h = 0.002
    # turns into different weights on those pixels: most pixels agree closely, the mean does)
    oref = O.nl_means(to_srgb(oimg), **FORK)
    close = np.abs(out - oref).max(2) <= 2e-2
    assert close.mean() > 0.9, close.mean()
    assert abs(out.mean() - oref.mean()) < 5e-3 * oref.mean()
    g.close(); hs.close()
