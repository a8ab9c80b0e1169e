"""Child process of tests/test_gpu_parity.py::test_rccl_gather_world_size_one: backend "nccl" (= RCCL) with ONE rank on the
GPU box, started clean (the process initialises the GPU itself; nothing is re-executed). Drives the product's own
collective helpers -- sharding.OverlappedGather.launch / wait (async_op=True on RCCL's stream) and sharding.all_gather_rows --
on DEVICE tensors around a real bootstrapped gate step, and checks the gathered block word for word."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch
import torch.distributed as dist


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    import oracle_lib as ol
    import redsec_amd
    from redsec_amd import sharding
    ks = ol.KeySet(ol.params("toy"), seed=3)
    ctx = ol.Ctx(ks)
    be = redsec_amd.Backend(redsec_amd.params("default128", n=ks.p.n), device=0)
    be.load_keys(ks.bk, ks.ksk)
    rng = np.random.default_rng(5)
    B = 37
    e8 = ol.to_torus(1, 8)
    ba, bb = rng.integers(0, 2, B), rng.integers(0, 2, B)
    ca = ks.encrypt(np.where(ba == 1, e8, -e8), 2.0 ** -15, 1)
    cb = ks.encrypt(np.where(bb == 1, e8, -e8), 2.0 ** -15, 2)
    da, db = torch.from_numpy(ca).cuda(), torch.from_numpy(cb).cuda()
    ref = ctx.gate_batch("NAND", ca, cb)
    pipe = sharding.OverlappedGather(B, be.W, torch.int32, torch.device("cuda", 0))
    assert not pipe.via_host and dist.get_backend() == "nccl"
    handles = []
    for step in range(3):                       # the collective of step k overlaps the kernels of step k + 1 (two buffers alternate)
        out = be.gate("NAND", da, db)
        h, full = pipe.launch(out)
        assert h is not None                    # async_op=True: a work handle on RCCL's stream
        handles.append((h, full))
        if len(handles) >= 2:
            hh, ff = handles[-2]
            pipe.wait(hh)
            assert np.array_equal(ff.cpu().numpy(), ref)
    pipe.wait(handles[-1][0])
    assert np.array_equal(handles[-1][1].cpu().numpy(), ref)
    got = sharding.all_gather_rows(be.gate("NAND", da, db), B, force=True)      # the ragged-slice collective, forced at one rank
    assert got.is_cuda and np.array_equal(got.cpu().numpy(), ref)
    dist.barrier()
    be.close()
    dist.destroy_process_group()
    print("rccl world-1 ok")


if __name__ == "__main__":
    main()
