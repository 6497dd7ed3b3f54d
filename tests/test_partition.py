"""The N > 1 path of bench.py on CPU: two gloo ranks render their spp slices (with the oracle standing in for the GPU,
per-sample streams) and all-reduce the films; the sum must be the single-process film of the same sample indices."""
import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-path-tracer_amd"))

from nori_hip.partition import spp_slice, split_spp, tile_bands  # noqa: E402


def test_slices_cover_the_sample_range():
    assert [spp_slice(r, 4, 16) for r in range(4)] == [(0, 16), (16, 32), (32, 48), (48, 64)]
    for total, world in [(64, 8), (10, 4), (3, 8), (0, 2)]:
        parts = split_spp(total, world)
        assert parts[0][0] == 0 and parts[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
        sizes = [e - b for b, e in parts]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        spp_slice(2, 2, 4)
    with pytest.raises(ValueError):
        spp_slice(0, 1, 0)


def test_tile_bands_partition_the_image():
    for height, world, band in [(600, 8, 32), (48, 2, 32), (31, 4, 32), (1080, 3, 32), (64, 8, 8)]:
        rows = np.zeros(height, np.int32)
        for r in range(world):
            for y0, y1 in tile_bands(r, world, height, band):
                assert 0 <= y0 < y1 <= height and y0 % band == 0
                rows[y0:y1] += 1
        assert np.all(rows == 1)                                   # every row exactly once
        sizes = [sum(y1 - y0 for y0, y1 in tile_bands(r, world, height, band)) for r in range(world)]
        assert max(sizes) - min(sizes) <= band
    assert tile_bands(3, 4, 31) == []                              # more ranks than bands: some ranks idle
    with pytest.raises(ValueError):
        tile_bands(4, 4, 100)


def test_group_plan_of_the_c_abi_tiles_samples_and_rows_for_1_to_8_members():
    """nori_hip_group_plan (include/nori_hip.h): the split nori_hip_group_render makes, as a pure function - no GPU is touched.
    For n = 1 .. 8 members: the spp slices tile the sample range in order, sizes within one of each other, empty for members
    beyond the sample count; the bands of 32 rows tile the image rows exactly once, member i holding bands i, i + n, ..."""
    import nori_hip as NH
    for n in range(1, 9):
        for b0, e0 in [(0, 16), (3, 4), (5, 5 + n), (0, 1), (7, 7 + 2 * n + 3), (100, 2148)]:
            parts = [NH.group_plan(n, i, b0, e0, 600, NH.PARTITION_SPP) for i in range(n)]
            assert all(p[2] == [] for p in parts)
            assert parts[0][0] == b0 and parts[-1][1] == e0
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))              # contiguous, in member order, disjoint
            sizes = [p[1] - p[0] for p in parts]
            assert min(sizes) >= 0 and max(sizes) - min(sizes) <= 1 and sum(sizes) == e0 - b0
            assert sizes == sorted(sizes, reverse=True)                              # the longer slices come first
            if e0 - b0 < n:
                assert sizes.count(0) == n - (e0 - b0)                              # spp < n: the last members idle
        for height in (1, 31, 32, 33, 600, 1080, 2160, 32 * n, 32 * n + 1):
            rows = np.zeros(height, np.int32)
            for i in range(n):
                sb, se, bands = NH.group_plan(n, i, 2, 9, height, NH.PARTITION_TILES)
                assert (sb, se) == (2, 9)                                           # every member renders all samples of its rows
                assert bands == [(y0, y1) for y0, y1 in tile_bands(i, n, height)]   # the same split bench.py's ranks use
                for k, (y0, y1) in enumerate(bands):
                    assert y0 == 32 * (i + k * n) and y1 == min(y0 + 32, height)
                    rows[y0:y1] += 1
            assert np.all(rows == 1)
    # the slices agree with bench.py's strong-scaling split of a sample count
    for total, n in [(64, 8), (10, 4), (3, 8)]:
        assert [NH.group_plan(n, i, 0, total, 64, NH.PARTITION_SPP)[:2] for i in range(n)] == split_spp(total, n)
    for bad in [(0, 0, 0, 4, 64, 0), (4, 4, 0, 4, 64, 0), (4, -1, 0, 4, 64, 0), (2, 0, 4, 4, 64, 0), (2, 0, 0, 4, 0, 1), (2, 0, 0, 4, 64, 7)]:
        with pytest.raises(NH.NoriHipError):
            NH.group_plan(*bad)


def _rank(rank, world, port, out):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from nori_hip.partition import spp_slice, allreduce_film
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    o = O.OracleScene(os.path.join(ROOT, "scenes/cbox/cbox_path_mis.xml"), 48, 40, 2)
    b, e = spp_slice(rank, world, 2)
    o.render(b, e, mode=1, threads=2)
    film = torch.from_numpy(np.ascontiguousarray(o.film()))
    allreduce_film(film, dist)
    dist.barrier()
    if rank == 0:
        np.save(out, film.numpy())
    dist.destroy_process_group()


def test_two_ranks_sum_to_the_single_process_film(tmp_path):
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    out = str(tmp_path / "film.npy")
    port = 29500 + os.getpid() % 2000
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_rank, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    got = np.load(out)
    o = O.OracleScene(os.path.join(ROOT, "scenes/cbox/cbox_path_mis.xml"), 48, 40, 4)
    o.render(0, 4, mode=1, threads=2)
    want = o.film()
    assert got.shape == want.shape
    # weights (sample counts) are integers and must match exactly; radiance sums differ by float addition order only
    assert np.array_equal(got[..., 3], want[..., 3])
    np.testing.assert_allclose(got[..., :3], want[..., :3], rtol=1e-5, atol=1e-6)


# ---- the tile-list path and configs[4]'s reduce-then-filter on two gloo ranks --------------------------------------------
W2, H2, SPP2 = 40, 72, 2            # three bands of 32 rows (the last one ragged): rank 0 gets two, rank 1 one
FILTER = dict(r=2, f=3, sigma=0.5, h=0.05)


def _to_srgb(v):
    v = np.asarray(v, np.float32)
    return np.where(v <= 0.0031308, 12.92 * v, 1.055 * np.power(np.maximum(v, 0), 1 / 2.4) - 0.055).astype(np.float32)   # src/common.cpp:192-206


def _filter_film(O, film, border):
    """what the filter pass does to a film: ImageBlock::toBitmap, sRGB, NL-means (cuda/filter.cc.cu:53-110)"""
    inner = film[border:film.shape[0] - border, border:film.shape[1] - border]
    w = inner[..., 3:4]
    rgb = np.where(w != 0, inner[..., :3] / np.where(w != 0, w, 1), 0).astype(np.float32)
    return O.nl_means(_to_srgb(rgb), FILTER["r"], FILTER["f"], FILTER["sigma"], FILTER["h"])


def _tile_film(O, o, tiles, spp):
    """the film of nori_hip_render_tiles(tiles, 0, spp) by the oracle: sample s of pixel (x, y) owns pcg32.seed(y W + x, s)"""
    px, py, st, sq = [], [], [], []
    for x0, y0, x1, y1 in tiles:
        ys, xs = np.mgrid[y0:y1, x0:x1]
        for s in range(spp):
            px.append(xs.ravel()); py.append(ys.ravel()); st.append(ys.ravel() * W2 + xs.ravel()); sq.append(np.full(xs.size, s))
    px, py, st, sq = (np.concatenate(a) for a in (px, py, st, sq))
    rgb, ps = o.li_samples(px, py, st.astype(np.uint64), sq.astype(np.uint64))
    return O.film_put(W2, H2, 2.0, ps, rgb)


def _rank_tiles(rank, world, port, out):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from nori_hip.partition import tile_bands, allreduce_film
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    o = O.OracleScene(os.path.join(ROOT, "scenes/cbox/cbox_path_mis.xml"), W2, H2, SPP2)
    tiles = [(0, y0, W2, y1) for y0, y1 in tile_bands(rank, world, H2)]           # this rank's bands: ONE tile-list call on the GPU
    film = torch.from_numpy(np.ascontiguousarray(_tile_film(O, o, tiles, SPP2)))
    allreduce_film(film, dist)
    dist.barrier()
    if rank == 0:
        # configs[4]: the filter pass runs once, after the reduce, on rank 0
        np.save(out, film.numpy())
        np.save(out + ".filtered.npy", _filter_film(O, film.numpy(), 2))
    dist.destroy_process_group()


def test_two_ranks_tile_lists_then_filter_match_the_single_process_result(tmp_path):
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    out = str(tmp_path / "film_tiles.npy")
    port = 31500 + os.getpid() % 2000
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_rank_tiles, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    got, got_f = np.load(out), np.load(out + ".filtered.npy")
    o = O.OracleScene(os.path.join(ROOT, "scenes/cbox/cbox_path_mis.xml"), W2, H2, SPP2)
    o.render(0, SPP2, mode=1, threads=2)                     # the whole image in one process
    want = o.film()
    assert np.array_equal(got[..., 3], want[..., 3])
    np.testing.assert_allclose(got[..., :3], want[..., :3], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(got_f, _filter_film(O, want, 2), rtol=1e-3, atol=1e-5)
