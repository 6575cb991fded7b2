"""CPU, world_size 2 over gloo: the N>1 path of the job -- contiguous sharding of independent
spectrograms, no collective in the update loop, one all-reduce of the residual pair at the end.
The per-shard "engine" here is the CPU checker (there is no GPU in this container); on GPUs the same
helpers are driven by bench.py with the HIP engine and the nccl (RCCL) backend."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from lws_amd.dist import shard_range, reduce_residual, gather_shards
    import lws_amd as L
    from oracle.oracle import Oracle
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B, T, F = 5, 8, 33
    rng = np.random.default_rng(123)
    S = rng.standard_normal((B, T, F)) + 1j * rng.standard_normal((B, T, F))
    p_awin = np.sqrt(L.hann(64))
    awin = np.sqrt(p_awin * L.synthwin(p_awin, 16))
    W = L.create_weights(awin, L.synthwin(awin, 16), 16, 5)
    thr = L.get_thresholds(3, 1.0, 0.1, 1)
    lo, hi = shard_range(B, rank, world)
    orc = Oracle()
    mine = np.stack([orc.batch_lws(S[b], W, thr) for b in range(lo, hi)])
    pairs = np.stack([[np.sum(np.abs(mine[i] - S[lo + i]) ** 2), np.sum(np.abs(mine[i]) ** 2)] for i in range(hi - lo)])
    err, pw, db = reduce_residual(pairs)
    full = gather_shards(mine, B)
    q.put((rank, lo, hi, err, pw, db, full))
    dist.destroy_process_group()


def test_shard_range_partitions():
    from lws_amd.dist import shard_range
    for n in (0, 1, 7, 8, 8192):
        for w in (1, 2, 3, 8):
            blocks = [shard_range(n, r, w) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def test_two_process_sharded_job_over_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    [p.join(timeout=60) for p in procs]
    assert [(r[1], r[2]) for r in res] == [(0, 3), (3, 5)]
    # both ranks hold the same global residual and the same gathered batch
    assert res[0][3] == res[1][3] and res[0][4] == res[1][4]
    assert np.array_equal(res[0][6], res[1][6]) and res[0][6].shape == (5, 8, 33)
    # and it equals the single-process answer
    sys.path.insert(0, ROOT)
    from oracle.oracle import Oracle
    import lws_amd as L
    rng = np.random.default_rng(123)
    S = rng.standard_normal((5, 8, 33)) + 1j * rng.standard_normal((5, 8, 33))
    a = np.sqrt(L.hann(64)); awin = np.sqrt(a * L.synthwin(a, 16))
    W = L.create_weights(awin, L.synthwin(awin, 16), 16, 5)
    thr = L.get_thresholds(3, 1.0, 0.1, 1)
    orc = Oracle()
    ref = np.stack([orc.batch_lws(S[b], W, thr) for b in range(5)])
    assert np.array_equal(res[0][6], ref)
    assert abs(res[0][3] - np.sum(np.abs(ref - S) ** 2)) < 1e-9 * res[0][3]
