"""Gate-parallel encrypted sign1024x1 over N GPUs (partitioning 2): ONE image, every
rank bootstraps 1/N of each stage's ciphertexts, RCCL all_gather before the next linear stage.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port 29544 \
      tools/mnist_gate_parallel.py [sign1024x1|sign1024x2|sign1024x3]

Checks on every rank that the sharded logits equal the unsharded ones word for word, prints the
latency of both. REDSEC_BENCH_REHEARSAL=1: ranks share device 0 and gather over gloo (one-GPU box).
"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import torch.distributed as dist
import redsec_amd
from redsec_amd import client, nets
import plain_model as pm

name = sys.argv[1] if len(sys.argv) > 1 else "sign1024x1"
rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
rehearsal = os.environ.get("REDSEC_BENCH_REHEARSAL") == "1"
gpu = 0 if rehearsal else int(os.environ.get("LOCAL_RANK", "0"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(gpu)
if rehearsal:
    dist.init_process_group("gloo", rank=rank, world_size=world)
else:
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", gpu))
sk = client.SecretKeySet("redsec_small_v2", seed=7)            # same key on every rank (seeded)
be = redsec_amd.Backend(redsec_amd.params("redsec_small_v2"), gpu)
be.load_keys(sk.bk, sk.ksk)
enc = nets.EncryptedMnist(be, pm.load_net(name))
labels, pixels = pm.load_images()
ct = torch.from_numpy(sk.encrypt_image(pixels[1], seed=5)).cuda(gpu)


def timed(shard):
    for _ in range(2):
        out = enc.run(ct, shard=shard)
    torch.cuda.synchronize(); dist.barrier()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); out = enc.run(ct, shard=shard); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    t = torch.tensor([min(ts)], dtype=torch.float64, device="cpu" if rehearsal else ct.device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return out, float(t.item())


ref, t_one = timed(False)
got, t_shard = timed(True)
same = torch.tensor([1 if torch.equal(ref, got) else 0], device="cpu" if rehearsal else ct.device)
dist.all_reduce(same, op=dist.ReduceOp.MIN)
if rank == 0:
    logits = sk.decrypt_ints(got.cpu().numpy())
    print(json.dumps({"net": name, "n_gpus": world, "unsharded_ms": round(t_one * 1e3, 2), "gate_parallel_ms": round(t_shard * 1e3, 2),
                      "sharded_equals_unsharded_word_for_word_on_every_rank": bool(same.item()),
                      "label": int(labels[1]), "encrypted_argmax": int(np.argmax(logits)),
                      "collective": "all_gather after each bootstrap stage (%s)" % ("gloo rehearsal, ranks share one GPU" if rehearsal else "RCCL")}))
dist.barrier()
dist.destroy_process_group()
