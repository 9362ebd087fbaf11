"""CPU, world_size 2 over gloo: the N>1 path -- contiguous image shards, the
fixed-layout detection gather to rank 0, max-over-ranks timing reduction."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dafne_amd.evaluation.gather import gather_detections, shard_range, to_predictions
    n_total, k_cap = 6, 32
    lo, hi = shard_range(n_total, rank, world)
    b = hi - lo
    rows = torch.zeros(b, k_cap, 18)
    counts = torch.zeros(b, dtype=torch.int32)
    for j in range(b):
        g = lo + j                      # global image id
        counts[j] = g + 1
        rows[j, : g + 1, 8] = torch.arange(g + 1, 0, -1).float() / 10 + g   # scores tagged with image id
        rows[j, : g + 1, 10] = g
    out = gather_detections(rows, counts, dst=0)
    t = torch.tensor([0.5 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        ra, ca = out
        preds = to_predictions(ra, ca)
        ok = ca.tolist() == [1, 2, 3, 4, 5, 6] and all(
            int(p["labels"][0]) == i and len(p["scores"]) == i + 1 for i, p in enumerate(preds))
        q.put(("rank0", ok, float(t)))
    else:
        q.put(("rank1", out is None, float(t)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_and_timing_reduction():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res = dict((r[0], r[1:]) for r in res)
    assert res["rank0"][0] and res["rank1"][0]
    assert res["rank0"][1] == 1.5 and res["rank1"][1] == 1.5      # MAX over ranks


def test_shard_range_covers_everything_once():
    from dafne_amd.evaluation.gather import shard_range
    for n, w in ((64, 8), (7, 2), (3, 8), (0, 4)):
        seen = []
        for r in range(w):
            lo, hi = shard_range(n, r, w)
            seen += list(range(lo, hi))
        assert seen == list(range(n))
