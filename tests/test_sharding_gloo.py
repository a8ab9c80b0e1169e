"""N>1 path on CPU: world-size-2 gloo processes shard an independent-ciphertext stage and gather
the slices (redsec_amd/sharding.py); result must equal the unsharded stage."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from redsec_amd import sharding


def test_shard_range_is_a_balanced_partition():
    for total in (0, 1, 7, 8, 196, 1024, 65536, 65537):
        for world in (1, 2, 3, 8):
            spans = [sharding.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, total, W, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(1234)            # same batch on every rank
        batch = torch.randint(-2**31, 2**31 - 1, (total, W), dtype=torch.int64, generator=g).to(torch.int32)
        stage = lambda rows: rows * 3 + 7                  # stands in for a per-ciphertext bootstrap
        got = sharding.sharded_stage(stage, batch)
        ok = torch.equal(got, stage(batch))
        lo, hi = sharding.shard_range(total, rank, world)
        ok = ok and torch.equal(sharding.all_gather_rows(batch[lo:hi].contiguous(), total), batch)
        flag = torch.tensor([1 if ok else 0])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if rank == 0:
            ret.put(int(flag.item()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [10, 197])              # even and ragged splits
def test_sharded_stage_equals_unsharded_gloo(total):
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, 21, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret.get(timeout=5) == 1
