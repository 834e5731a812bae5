"""N>1 path on CPU: two gloo ranks each take a contiguous span of blocks (halo + global first-block index), process
it independently and rank 0 gathers — the same plan bench.py --gpus N uses with one GPU per rank.  On the CPU the span
is processed by the oracle (checker); the point of the test is the span/halo/phase bookkeeping and the
torch.distributed plumbing (barrier, MAX reduction, gather), which are identical on the GPU path."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    import gr_fdc_amd as G
    import oracle as O
    dist.init_process_group("gloo", rank=rank, world_size=world)
    N, R, total = 1024, 4, 11                      # 11 blocks over 2 ranks: ragged spans (6 + 5)
    H = N - N // R
    chans = [(33, 64, 0.6, 0.85), (513, 256, 0.88, 1.0), (7, 16, 0.5, 0.9)]     # odd f: phase depends on the global index
    rng = np.random.default_rng(5)
    x = (rng.standard_normal(total * H) + 1j * rng.standard_normal(total * H)).astype(np.complex64)
    first, n = G.span_for_rank(total, rank, world)
    ring = G.ring_for_span(x, first, n, N, R)
    assert ring.size == N // R + n * H
    outs, _ = O.channelizer(N, R, 1, chans, ring[N // R:], prefix=ring[:N // R], first_block=first)
    # timing plumbing as in bench.py: barrier, then MAX over ranks
    dist.barrier()
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.item() == world
    gathered = [None] * world
    dist.all_gather_object(gathered, (first, n, [o.copy() for o in outs]))
    if rank == 0:
        whole, _ = O.channelizer(N, R, 1, chans, x)
        gathered.sort(key=lambda g: g[0])
        assert [g[0] for g in gathered] == [0, 6] and sum(g[1] for g in gathered) == total
        for c in range(len(chans)):
            cat = np.concatenate([g[2][c] for g in gathered])
            assert (cat.view(np.uint32) == whole[c].view(np.uint32)).all()
        open(os.path.join(tmp, "ok"), "w").write("ok")
    dist.destroy_process_group()


def test_two_rank_span_sharding_equals_single(tmp_path):
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()


def test_span_for_rank_partitions():
    import gr_fdc_amd as G
    for total in (1, 7, 8, 1024, 1025):
        for world in (1, 2, 3, 8):
            spans = [G.span_for_rank(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(n for _f, n in spans) == total
            for (f0, n0), (f1, _n1) in zip(spans, spans[1:]):
                assert f1 == f0 + n0
    assert G.ring_bounds(0, 4, 1024, 4) == (-256, 4 * 768)
