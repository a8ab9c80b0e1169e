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


def test_device_to_device_exchange_schedule():
    """The copy plan of rs_allgather_rows (csrc/rs_host.h::exchange_schedule, the C++ mirror's multi-device slice exchange): every
    destination pulls every other rank's slice exactly once, the slices are sharding.shard_range's, and the n copies of a
    round have n different sources and n different destinations (xGMI is point to point: no link carries two at a time)."""
    import ctypes as C
    import emu_lib
    L = emu_lib.lib()
    for rows in (0, 1, 5, 10, 196, 1024, 131072, 131075):
        for n in (1, 2, 3, 4, 8):
            cnt = L.rs_emu_exchange_schedule(rows, n, None)
            buf = (C.c_long * (5 * max(cnt, 1)))()
            assert L.rs_emu_exchange_schedule(rows, n, buf) == cnt
            plan = [tuple(buf[5 * i:5 * i + 5]) for i in range(cnt)]
            spans = [sharding.shard_range(rows, r, n) for r in range(n)]
            want = {(d, e) for d in range(n) for e in range(n) if d != e and spans[e][1] > spans[e][0]}
            assert {(d, e) for d, e, _, _, _ in plan} == want and len(plan) == len(want)
            for d, e, k, lo, hi in plan:
                assert (lo, hi) == spans[e] and 1 <= k < n and e == (d + k) % n
            for k in range(1, n):
                rnd = [(d, e) for d, e, kk, _, _ in plan if kk == k]
                assert len({d for d, _ in rnd}) == len(rnd) == len({e for _, e in rnd})


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


def _pipe_worker(rank, world, port, rows, W, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pipe = sharding.OverlappedGather(rows, W, torch.int32, "cpu")
        ok = True
        pending = []
        for k in range(4):                                   # bench.py's loop: two buffers alternate
            if len(pending) >= 2:
                pipe.wait(pending[-2][0])
            local = torch.full((rows, W), 100 * k + rank, dtype=torch.int32)
            pending.append(pipe.launch(local))
        for k, (h, full) in enumerate(pending[-2:], start=2):
            pipe.wait(h)
            want = torch.cat([torch.full((rows, W), 100 * k + r, dtype=torch.int32) for r in range(world)])
            ok = ok and torch.equal(full, want)
        flag = torch.tensor([1 if ok else 0])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if rank == 0:
            ret.put(int(flag.item()))
    finally:
        dist.destroy_process_group()


def test_overlapped_gather_pipeline_gloo():
    """The collective of bench.py --gpus N (sharding.OverlappedGather) on two gloo ranks."""
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipe_worker, args=(r, 2, port, 5, 7, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret.get(timeout=5) == 1


def _gpu_stage_worker(rank, world, port, ret):
    """Two ranks sharing device 0 (gloo: RCCL refuses two ranks on one device) shard a REAL bootstrap stage."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
        import redsec_amd
        from redsec_amd import client
        sk = client.SecretKeySet("redsec_small_v2", seed=3, n=24)            # same key on both ranks (seeded)
        be = redsec_amd.Backend(redsec_amd.params("redsec_small_v2", n=24), 0)
        be.load_keys(sk.bk, sk.ksk)
        B = 37                                                              # ragged split: 19 + 18
        ct = torch.from_numpy(sk.encrypt_torus(np.arange(B) * (1 << 24) - (1 << 28), seed=5)).cuda()
        mu = 1 << 20
        whole = be.bootstrap(ct, mu)
        got = sharding.sharded_stage(lambda rows: be.bootstrap(rows, mu), ct)
        ok = torch.equal(got, whole)
        lo, hi = sharding.shard_range(B, rank, world)
        pipe = sharding.OverlappedGather(19, be.W, torch.int32, "cuda:0")    # equal blocks: pad the shorter slice
        block = torch.zeros((19, be.W), dtype=torch.int32, device="cuda")
        block[: hi - lo] = be.bootstrap(ct[lo:hi].contiguous(), mu)
        h, full = pipe.launch(block)
        pipe.wait(h)
        full = full.reshape(world, 19, be.W)
        ok = ok and torch.equal(torch.cat([full[0, :19], full[1, :18]]).cuda(), whole)
        flag = torch.tensor([1 if ok else 0])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if rank == 0:
            ret.put(int(flag.item()))
        be.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_sharded_bootstrap_stage_two_ranks_one_device():
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_stage_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    assert ret.get(timeout=5) == 1


def test_exchange_plan_orders_every_copy_and_falls_back_to_host_staging():
    """The literal operation list rs_allgather_rows issues (csrc/rs_host.h::exchange_plan), for device sets with and without peer
    access: (1) a destination's copy stream waits for the destination's OWN "slice" event before its first copy -- the block it
    receives into may have been recycled while kernels on its compute stream still read it (round-3 advisor finding);
    (2) every copy is preceded, on the same stream, by a wait for its source (the source's "slice" event on the direct paths,
    its "staged" event on the host path); (3) the path is same-device / peer / host-staged exactly as hipDeviceCanAccessPeer
    allows, and forcing host staging covers every pair; (4) a source stages once, behind its own slice and behind every
    context's previous "copied" event (its pinned buffer may still be read), before anyone waits for it."""
    import ctypes as C
    import itertools
    import emu_lib
    L = emu_lib.lib()
    L.rs_emu_exchange_plan.restype = C.c_long
    WAIT_SLICE, WAIT_COPIED, STAGE_OUT, WAIT_STAGED, COPY, RECORD_COPIED = range(6)
    SAME, PEER, STAGED = range(3)
    cases = [([0, 0, 0], None, 0), ([0, 1, 2, 3], "all", 0), ([0, 1, 2, 3], "none", 0), ([0, 1, 2], "ring", 0), ([0, 0, 1, 1], "none", 0),
             ([0, 0], None, 1), ([0, 1, 2, 3, 4, 5, 6, 7], "all", 0), ([0, 1, 2, 3, 4, 5, 6, 7], "all", 1)]
    for devices, access, force in cases:
        n = len(devices)
        peer = [[1] * n for _ in range(n)]
        if access == "none":
            peer = [[1 if devices[d] == devices[s] else 0 for s in range(n)] for d in range(n)]
        elif access == "ring":                                     # only neighbours see each other
            peer = [[1 if abs(d - s) in (0, 1, n - 1) and not (n == 3 and {d, s} == {0, 2}) else 0 for s in range(n)] for d in range(n)]
        flat = (C.c_ubyte * (n * n))(*itertools.chain.from_iterable(peer))
        dev = (C.c_int * n)(*devices)
        for rows in (5, 196, 131072):
            cnt = L.rs_emu_exchange_plan(rows, n, dev, flat, force, None)
            buf = (C.c_long * (6 * cnt))()
            assert L.rs_emu_exchange_plan(rows, n, dev, flat, force, buf) == cnt
            ops = [tuple(buf[6 * i:6 * i + 6]) for i in range(cnt)]
            spans = [sharding.shard_range(rows, r, n) for r in range(n)]
            waited = {c: set() for c in range(n)}                  # per copy stream: ("slice" | "staged" | "copied", ctx) seen so far
            staged_at, copies = {}, set()
            for pos, (kind, ctx, other, path, lo, hi) in enumerate(ops):
                if kind == WAIT_SLICE:
                    waited[ctx].add(("slice", other))
                elif kind == WAIT_COPIED:
                    waited[ctx].add(("copied", other))
                elif kind == WAIT_STAGED:
                    assert other in staged_at, "waiting for a staging copy that was never issued"
                    waited[ctx].add(("staged", other))
                elif kind == STAGE_OUT:
                    assert ctx not in staged_at and (lo, hi) == spans[ctx]
                    assert ("slice", ctx) in waited[ctx] and all(("copied", e) in waited[ctx] for e in range(n))
                    staged_at[ctx] = pos
                elif kind == COPY:
                    assert ("slice", ctx) in waited[ctx], "destination did not wait for its own compute stream"
                    want = STAGED if force else (SAME if devices[ctx] == devices[other] else (PEER if peer[ctx][other] else STAGED))
                    assert path == want and (lo, hi) == spans[other] and hi > lo
                    assert (("staged", other) if path == STAGED else ("slice", other)) in waited[ctx]
                    copies.add((ctx, other))
                elif kind == RECORD_COPIED:
                    assert all((ctx, e) in copies for e in range(n) if e != ctx and spans[e][1] > spans[e][0])
            assert copies == {(d, e) for d in range(n) for e in range(n) if d != e and spans[e][1] > spans[e][0]}
            assert sum(1 for o in ops if o[0] == RECORD_COPIED) == n


def _image_worker(rank, world, port, n_images, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        images = [torch.full((10, 5), 7 * i + 1, dtype=torch.int32) for i in range(n_images)]
        ran = []

        def run_image(x):
            ran.append(int(x[0, 0]))
            return x * 3 - 2                                # stands in for a whole encrypted network
        got, t_c, t_g = sharding.image_parallel(run_image, images, (10, 5), device="cpu")
        want = torch.stack([x * 3 - 2 for x in images]) if images else torch.empty((0, 10, 5), dtype=torch.int32)
        ok = torch.equal(got, want) and got.device.type == "cpu"
        ok = ok and ran == [7 * i + 1 for i in sharding.image_assignment(n_images, rank, world)] and t_c >= 0 and t_g >= 0
        flag = torch.tensor([1 if ok else 0])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if rank == 0:
            ret.put(int(flag.item()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_images", [2, 5, 1, 0])        # one each (configs[4]'s shape), a ragged batch, a rank without an image, nobody has one
def test_image_parallel_replicas_gloo(n_images):
    """BASELINE configs[4] on the CPU: image-parallel replicas over two gloo ranks, the logits gathered in image order."""
    assert sharding.image_assignment(8, 3, 8) == [3] and sharding.image_assignment(5, 1, 2) == [1, 3]
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_image_worker, args=(r, 2, port, n_images, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret.get(timeout=5) == 1


def test_image_parallel_without_a_process_group_is_the_plain_loop():
    images = [torch.full((10, 3), i, dtype=torch.int32) for i in range(3)]
    got, _, t_g = sharding.image_parallel(lambda x: x + 1, images, (10, 3))
    assert torch.equal(got, torch.stack(images) + 1) and t_g == 0.0
    # an empty batch: an empty [0][classes][W] tensor on the named device, no collective (also with one rank and `force`)
    for force in (False, True):
        got, _, t_g = sharding.image_parallel(lambda x: x + 1, [], (10, 3), force=force, device="cpu")
        assert got.shape == (0, 10, 3) and got.dtype == torch.int32 and got.device.type == "cpu" and t_g == 0.0
