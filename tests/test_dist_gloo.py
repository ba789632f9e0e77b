"""The N>1 plumbing of bench.py (otezip_amd/dist.py) on CPU: two processes, gloo backend.
Entries shard without a data-path collective; only the timing/counter reductions communicate."""
import os
import socket

import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from otezip_amd import dist as D
    r, lr, w = D.init("gloo")
    lo, hi = D.shard_range(10001, w, r)
    D.barrier()
    elapsed = D.reduce_max(1.0 + r)          # the slowest rank defines the step time
    total = D.reduce_sum(hi - lo)            # every entry is owned exactly once
    D.barrier()
    D.finish()
    q.put((r, lo, hi, elapsed, total))


def test_two_ranks_shard_and_reduce():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    got = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [g[1:3] for g in got] == [(0, 5001), (5001, 10001)]
    assert all(g[3] == 2.0 and g[4] == 10001 for g in got)


def test_shard_range_properties():
    from otezip_amd.dist import shard_range
    for n in (0, 1, 7, 10000, 65535):
        for w in (1, 2, 3, 4, 8):
            spans = [shard_range(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
