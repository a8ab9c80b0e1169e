"""Rank process of tests/test_gpu_cifar.py::test_image_parallel_two_ranks_one_device (BASELINE configs[4] rehearsed on a
one-GPU box): started by torch.distributed.run with REDSEC_BENCH_REHEARSAL=1, so the ranks share device 0 and talk over
gloo (RCCL refuses two ranks on one device). Every rank builds the same seeded key, runs ITS images of the batch through
nets.EncryptedCifar via sharding.image_parallel -- the path bench.py's cifar_batch leg times -- and rank 0 writes the
gathered logit ciphertexts [n_images][10][W] to the .npy path given as argv[1]. argv[2]: net name, argv[3]: image count."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch
import torch.distributed as dist


def main():
    out_path, net_name, n_images = sys.argv[1], sys.argv[2], int(sys.argv[3])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert os.environ.get("REDSEC_BENCH_REHEARSAL") == "1"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import plain_model as pm
    import redsec_amd
    from redsec_amd import client, nets, sharding
    sk = client.SecretKeySet("redsec_small_v2", seed=19)
    be = redsec_amd.Backend(redsec_amd.params("redsec_small_v2"), 0)
    be.load_keys(sk.bk, sk.ksk)
    enc = nets.EncryptedCifar(be, pm.CifarNet(net_name))
    _, pix = pm.load_cifar_images()
    images = [(k + 1) % len(pix) for k in range(n_images)]
    ran = []

    def run_image(i):
        ran.append(i)
        return enc.run(torch.from_numpy(sk.encrypt_image(pix[i], seed=100 + i)).cuda())
    logits, t_compute, t_gather = sharding.image_parallel(run_image, images, (10, be.W))
    assert ran == [images[k] for k in sharding.image_assignment(n_images, rank, world)]
    assert be.rounding_certificate() < 0.2 and be.fft_fallbacks() == 0
    every = [None] * world
    dist.all_gather_object(every, logits.cpu().numpy().tobytes())
    assert all(e == every[0] for e in every)                    # every rank holds the same gathered batch
    if rank == 0:
        np.save(out_path, logits.cpu().numpy())
    dist.barrier()
    be.close()
    dist.destroy_process_group()
    print("rank %d ok: images %s, compute %.2f s, gather %.4f s" % (rank, ran, t_compute, t_gather))


if __name__ == "__main__":
    main()
