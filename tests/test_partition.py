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
