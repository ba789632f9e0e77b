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


def _table_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import numpy as np
    from otezip_amd import corpus, dist as D
    D.init("gloo")
    table = None
    if rank == 0:
        # what a host does after zip_open: the decode-relevant directory fields of every entry
        # (entry sizes vary 4x here, unlike the benchmark corpus, so that balance is not trivial)
        rows = []
        for i in range(400):
            arc, ents = corpus.build_archive(1, 4096 + 3 * 4096 * (i % 4), workers=1, first=i)
            e = ents[0]
            rows.append((e["method"], e["data_off"] + 100000 * i, e["comp_size"], e["uncomp_size"], e["crc32"]))
        table = np.array(rows, dtype=np.int64)
    table = D.broadcast_table(table, src=0)
    spans = D.partition_by_bytes(table[:, 2] + table[:, 3], world)
    lo, hi = spans[rank]
    mine = int((table[lo:hi, 2] + table[lo:hi, 3]).sum())
    total = D.reduce_sum(mine)
    n_owned = D.reduce_sum(hi - lo)
    D.barrier()
    D.finish()
    q.put((rank, lo, hi, mine, total, n_owned, int(table[:, 4].sum()), table.shape))


def test_descriptor_table_broadcast_and_byte_balanced_split():
    """The work queue of the multi-GPU path: rank 0's descriptor table reaches every rank intact, the
    ranks' runs are a disjoint exact cover in index order, and their byte counts differ by < 5 %."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_table_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    got = sorted(q.get(timeout=300) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, lo0, hi0, b0, tot0, n0, crc0, shape0), (r1, lo1, hi1, b1, tot1, n1, crc1, shape1) = got
    assert shape0 == shape1 == (400, 5) and crc0 == crc1          # same table everywhere
    assert lo0 == 0 and hi0 == lo1 and hi1 == 400 and n0 == n1 == 400  # disjoint exact cover
    assert tot0 == tot1 == b0 + b1
    assert abs(b0 - b1) / (b0 + b1) < 0.05


def test_partition_by_bytes_properties():
    import numpy as np
    from otezip_amd.dist import partition_by_bytes
    rng = np.random.default_rng(7)
    for n in (0, 1, 5, 1000, 10000):
        for w in (1, 2, 3, 8):
            costs = rng.integers(1, 200000, size=n)
            spans = partition_by_bytes(costs, w)
            assert len(spans) == w and spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] and a[0] <= a[1] for a, b in zip(spans, spans[1:]))
            if n >= 1000:
                sums = [int(costs[a:b].sum()) for a, b in spans]
                assert (max(sums) - min(sums)) / (sum(sums) / w) < 0.05


def test_shard_range_properties():
    from otezip_amd.dist import shard_range
    for n in (0, 1, 7, 10000, 65535):
        for w in (1, 2, 3, 4, 8):
            spans = [shard_range(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
