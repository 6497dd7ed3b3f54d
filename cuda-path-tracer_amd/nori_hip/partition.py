"""How a render is split over GPUs (one process per GPU): every rank renders the same image with its own slice of the
sample indices and the fp32 films (rgb sums + weight per pixel) are added.  Per-sample pcg32 streams are keyed by
(pixel, sample index), so a slice renders exactly the samples a single process would render for those indices and
the summed film differs from the single-process one only by the order of float additions.

The reference has no multi-GPU path (it renders 32x32 blocks on TBB threads, src/render.cpp:72-101); the split follows
its block merge ImageBlock::put(ImageBlock&) (src/block.cpp:131-140), which is also a plain sum of films.
"""


def spp_slice(rank, world, spp_per_rank):
    """sample-index range [begin, end) of `rank` when every rank renders spp_per_rank samples per pixel (weak scaling)"""
    if world < 1 or not (0 <= rank < world) or spp_per_rank < 1:
        raise ValueError("bad partition: rank %r of %r, %r spp" % (rank, world, spp_per_rank))
    return rank * spp_per_rank, (rank + 1) * spp_per_rank


def split_spp(total_spp, world):
    """strong-scaling split of a fixed sample count: contiguous ranges whose sizes differ by at most one; ranks beyond
    the sample count get an empty range"""
    if world < 1 or total_spp < 0:
        raise ValueError("bad partition")
    base, rem = divmod(total_spp, world)
    out, b = [], 0
    for r in range(world):
        n = base + (1 if r < rem else 0)
        out.append((b, b + n))
        b += n
    return out


def tile_bands(rank, world, height, band=32):
    """strong-scaling split by image tiles (SURVEY §8e ii): the rows of `rank` as [y0, y1) bands of `band` rows (the
    reference's block size, NORI_BLOCK_SIZE), dealt round-robin so that every rank gets rows from all over the image.
    Bands of different ranks are disjoint, so summing the films (the same all-reduce as for spp slices) assembles the
    image; every rank renders all samples of its pixels."""
    if world < 1 or not (0 <= rank < world) or height < 1 or band < 1:
        raise ValueError("bad partition: rank %r of %r, height %r, band %r" % (rank, world, height, band))
    n_bands = (height + band - 1) // band
    return [(k * band, min(height, (k + 1) * band)) for k in range(rank, n_bands, world)]


def allreduce_film(film, dist):
    """sum the ranks' films in place (torch tensor, any device); a no-op without a process group.  The caller has
    synchronised the render stream before (the collective runs on torch's stream and reads the film); on return the
    collective has completed, so the render stream may clear or extend the film again."""
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(film)
        if film.is_cuda:
            import torch
            torch.cuda.current_stream(film.device).synchronize()
    return film
