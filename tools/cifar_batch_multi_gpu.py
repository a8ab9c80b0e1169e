"""BASELINE configs[4]: a batch of encrypted CIFAR images, ONE IMAGE PER GPU, full key replica per GPU,
the logit ciphertexts gathered at the end (the only exchange step of image-parallel replicas,
SURVEY.md section 8e). One process per GPU:

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29533 \
      tools/cifar_batch_multi_gpu.py [binarynet|binarynet_small]

Rank r encrypts image r with the reference's own client tool, runs the reference's UNMODIFIED
nets/cifar/<net>/{net,main}.cpp driver (built against redsec_amd/host by redsec_amd/build.py) on GPU r
as a child process, reads the TFHE-format result file and contributes its 10 x (n+1) words to an
all_gather over RCCL; rank 0 decrypts all of them. With WORLD_SIZE=1 it is the single-GPU run.
REDSEC_BENCH_REHEARSAL=1 walks the same path on a one-GPU box (ranks share device 0, gloo).
"""
import json, os, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import torch.distributed as dist
import plain_model as pm, refdrivers as rd

net_name = sys.argv[1] if len(sys.argv) > 1 else "binarynet"
rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
local_rank = int(os.environ.get("LOCAL_RANK", "0"))
rehearsal = os.environ.get("REDSEC_BENCH_REHEARSAL") == "1"
gpu = 0 if rehearsal else local_rank
if world > 1:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if rehearsal:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        torch.cuda.set_device(gpu)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", gpu))

# ---- keys: generated once (rank 0) into a directory every rank of the node can read ----
shared = os.environ.get("REDSEC_SHARED_DIR") or os.path.join(tempfile.gettempdir(), "redsec_cifar_batch_%s" % os.environ.get("MASTER_PORT", "0"))
keydir = os.path.join(shared, "keys")
if rank == 0:
    shutil.rmtree(shared, ignore_errors=True); os.makedirs(keydir)
    assert rd.run("client_gen_secure_keyset.out", keydir).returncode == 0
if world > 1:
    dist.barrier()

# ---- this rank's scratch tree: client/ (keys linked, its own image) + nets/cifar/<net>/ ----
tree = os.path.join(shared, "rank%d" % rank)
client = os.path.join(tree, "client"); netdir = os.path.join(tree, "nets", "cifar", net_name)
os.makedirs(client); os.makedirs(netdir)
for f in os.listdir(keydir):
    os.symlink(os.path.join(keydir, f), os.path.join(client, f))
shutil.copyfile(os.path.join(rd.GOLD, "cifar_%s_var_prep.dat" % net_name), os.path.join(netdir, "var_prep.dat"))
labels, pix = pm.load_cifar_images()
img = (rank + 1) % len(labels)      # image 0 of the fixture set is misclassified by the plaintext net too
with open(os.path.join(client, "img.csv"), "w") as f:
    f.write(",".join(str(int(v)) for v in [labels[img], 32, 32, 3] + list(pix[img])) + ",\n")
assert rd.run("client_encrypt_image.out", client, "img.csv").returncode == 0

# ---- encrypted inference on this rank's GPU (child process: the reference's own driver) ----
os.environ["HIP_VISIBLE_DEVICES"] = str(gpu)
if world > 1:
    dist.barrier()
t0 = time.perf_counter()
r = rd.run("cifar_%s_enc.out" % net_name, netdir)
assert r.returncode == 0 and "Result ctxts loaded" in r.stdout, r.stdout[-400:] + r.stderr[-400:]
mine = torch.from_numpy(rd.read_ciphertexts(os.path.join(client, "network_output.ctxt"), 350, 10).copy())

# ---- the gather: 10 x 351 int32 per rank ----
if world > 1:
    if not rehearsal:
        del os.environ["HIP_VISIBLE_DEVICES"]
        mine = mine.cuda(gpu)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    allct = torch.stack([p.cpu() for p in parts]).numpy()
    tmax = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=mine.device)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    wall = float(tmax.item())
else:
    allct = mine.numpy()[None]
    wall = time.perf_counter() - t0

if rank == 0:
    _, lwe_key = rd.read_secret_key(os.path.join(keydir, "secret.key"))
    res = []
    for g in range(world):
        ct = allct[g]
        phase = (ct[:, 350].astype(np.int64) - (ct[:, :350].astype(np.int64) * lwe_key).sum(axis=1)) & 0xFFFFFFFF
        dec = ((phase + (1 << 19)) >> 20) & 0xFFF
        dec = np.where(dec > 2048, dec - 4096, dec)
        i = (g + 1) % len(labels)
        res.append({"image": int(i), "label": int(labels[i]), "encrypted_argmax": int(np.argmax(dec)),
                    "plaintext_argmax": int(np.argmax(pm.cifar_forward(pm.CifarNet(net_name), pix[i])))})
    print(json.dumps({"workload": "cifar/%s, %d encrypted image(s), one per GPU, logits gathered" % (net_name, world),
                      "n_gpus": world, "wall_s_max_over_ranks": round(wall, 3), "images_per_s": round(world / wall, 3),
                      "collective": "all_gather of 10 x 351 int32 per rank (%s)" % ("none: one rank" if world == 1 else "gloo rehearsal" if rehearsal else "RCCL"),
                      "results": res}))
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
if rank == 0:
    shutil.rmtree(shared, ignore_errors=True)
