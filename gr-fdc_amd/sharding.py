"""Block-span sharding of the overlap-save stream across GPUs.

The chain's only state is the N/R-sample history (lib/overlap_save_impl.h:33) and the per-channel window counter, which
is (m * shift) mod R in closed form (lib/phase_shifting_windowing_vcc_impl.cc:82).  A contiguous span of blocks
[first, first+n) is therefore independent of every other span given its N/R-sample halo and `first`: no collective on
the data path, only a host scatter/gather (or one point-to-point halo copy)."""


def span_for_rank(total_blocks, rank, world):
    """Contiguous, balanced spans: returns (first_block, nblocks) of `rank`."""
    base, extra = divmod(int(total_blocks), int(world))
    n = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, n


def ring_bounds(first_block, nblocks, N, R):
    """Sample range [lo, hi) of the stream (stream sample 0 = first new sample of block 0) that a span needs,
    halo included; lo is negative for the first span (zero history, lib/overlap_save_impl.cc:52)."""
    ovl = N // R
    H = N - ovl
    return first_block * H - ovl, (first_block + nblocks) * H


def ring_for_span(stream, first_block, nblocks, N, R):
    """numpy helper: the ring (halo + new samples) of a span cut out of the whole stream."""
    import numpy as np
    lo, hi = ring_bounds(first_block, nblocks, N, R)
    if lo < 0:
        return np.concatenate([np.zeros(-lo, dtype=stream.dtype), stream[:hi]])
    return stream[lo:hi]
