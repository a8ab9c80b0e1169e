"""Multi-GPU sharding of an independent-ciphertext stage (one process per GPU).

Every bootstrap of a layer stage is independent of every other one (the reference's loop at
lib/BinFunc.cpp:1056-1071 has no cross-iteration dependence), so ranks take contiguous slices of the
batch with full key replicas and no data-path collective; the only exchange is the gather of the
slices before the next linear stage (which needs the whole bit vector) or of the logits of
image-parallel replicas. Works on any torch.distributed backend: "nccl" (= RCCL over xGMI) on the
GPUs, "gloo" in the CPU tests.
"""
import torch
import torch.distributed as dist


def shard_range(total, rank, world):
    """Contiguous, balanced slice [lo, hi) of `total` rows for `rank`; sizes differ by at most 1."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_rows(local, total, group=None, force=False):
    """Concatenate per-rank row slices (made with shard_range) into the full [total][...] tensor on
    every rank. Ragged slices are padded to the largest one for the collective. A single rank returns its slice as it
    is unless `force` asks for the collective anyway (tests: the RCCL path on a one-GPU box)."""
    world = dist.get_world_size(group)
    if world == 1 and not force:
        return local
    sizes = [shard_range(total, r, world)[1] - shard_range(total, r, world)[0] for r in range(world)]
    width = max(sizes)
    # gloo (CPU tests, one-GPU rehearsals) moves host tensors; nccl (= RCCL over xGMI) device tensors
    via_host = local.is_cuda and dist.get_backend(group) == "gloo"
    src = local.cpu() if via_host else local
    pad = torch.zeros((width,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    pad[: src.shape[0]] = src
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    out = torch.cat([b[:s] for b, s in zip(bufs, sizes)], dim=0)
    return out.to(local.device) if via_host else out


def sharded_stage(fn, batch, group=None):
    """Run `fn(rows)` on this rank's slice of `batch` ([B][...]) and return the gathered [B][...]."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_range(batch.shape[0], rank, world)
    out = fn(batch[lo:hi].contiguous())
    return all_gather_rows(out, batch.shape[0], group) if world > 1 else out


class OverlappedGather:
    """All-gather of equally sized per-rank row blocks that overlaps with the NEXT step's kernels.

    The gate batch of a layer stage shards across the GPUs of a node with no exchange until its outputs are needed
    whole ("final RCCL gather over xGMI" of the north star). With backend "nccl" (= RCCL) the collective runs on the
    process group's own stream: launch() makes that stream wait for the kernels already enqueued on the current
    stream and returns at once; wait() makes the current stream wait for the collective. Two result buffers
    alternate, so step k+1 computes while the outputs of step k travel. With "gloo" (CPU tests, one-GPU
    rehearsals) the same calls run synchronously through host memory.
    """

    def __init__(self, rows_per_rank, width, dtype, device, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rows = int(rows_per_rank)
        self.via_host = torch.device(device).type == "cuda" and dist.get_backend(group) == "gloo"
        buf_dev = "cpu" if self.via_host else device
        self.full = [torch.empty((self.world * self.rows, width), dtype=dtype, device=buf_dev) for _ in range(2)]
        self.device = device
        self.k = 0

    def launch(self, local):
        """Start gathering `local` ([rows_per_rank][width]); returns (handle, full) where `full` is valid after wait(handle)."""
        assert local.shape[0] == self.rows
        full = self.full[self.k % 2]
        self.k += 1
        if self.via_host:
            dist.all_gather_into_tensor(full, local.cpu(), group=self.group)
            return None, full
        work = dist.all_gather_into_tensor(full, local, group=self.group, async_op=True)
        return work, full

    @staticmethod
    def wait(handle):
        if handle is not None:
            handle.wait()


def image_assignment(n_images, rank, world):
    """Images of an image-parallel batch this rank runs: r, r + world, r + 2 world ... (BASELINE configs[4]: 8 images on
    8 GPUs = one each). The reference's shape is enc_segs[NUM_GPUS], one host thread per GPU and no merge step
    (lib/GPU/Layer.cuh:15,22-37, nets/mnist/sign1024x1/main.cu:81-83); here the ranks are processes and the merge is the
    gather below."""
    return list(range(int(rank), int(n_images), int(world)))


def image_parallel(run_image, images, out_shape, group=None, force=False, device=None):
    """Image-parallel replicas (partitioning 1): every rank holds a full key replica and runs
    `run_image(images[i])` -> int32 [classes][W] (the logit ciphertexts, on the rank's device) for the images
    image_assignment gives it; the only exchange is ONE all-gather of classes x W words per image at the end (10 x 351 words
    = 14 KB for a CIFAR image: latency-bound, which is why it is a single collective). Returns (logits int32
    [n_images][classes][W] in image order, identical on every rank; this rank's seconds of compute before the collective;
    seconds in the collective). `images`: a sequence every rank can index (a rank touches only its own); `out_shape` =
    (classes, W), known to a rank even when it has no image of a short batch. Without an initialised process group (or one
    rank and not `force`) it is the plain loop. `device`: where a rank WITHOUT an image of a short batch builds its (all-zero)
    contribution and returns the result -- it must be the device the other ranks' logits live on (with the gloo backend on a
    GPU box that may be the CPU); default: the current HIP device if there is one. An empty batch returns an empty
    [0][classes][W] tensor and runs no collective."""
    import time
    n = len(images)
    on = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if on else 1
    rank = dist.get_rank(group) if on else 0
    mine = image_assignment(n, rank, world)
    t0 = time.perf_counter()
    outs = [run_image(images[i]) for i in mine]
    if outs and outs[0].is_cuda:
        torch.cuda.synchronize(outs[0].device)
    t_compute = time.perf_counter() - t0
    shape = tuple(int(v) for v in out_shape)
    if outs:
        dev = outs[0].device
    elif device is not None:
        dev = torch.device(device)
    else:
        dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    if n == 0:                                                  # every rank sees the same n: nobody enters the collective
        return torch.empty((0,) + shape, dtype=torch.int32, device=dev), t_compute, 0.0
    if world == 1 and not force:
        return torch.stack(outs), t_compute, 0.0
    t1 = time.perf_counter()
    per_rank = -(-n // world)                                   # slots per rank; a short rank pads with zeros
    assert all(tuple(o.shape) == shape for o in outs)
    via_host = dev.type == "cuda" and dist.get_backend(group) == "gloo"
    buf_dev = torch.device("cpu") if via_host else dev
    local = torch.zeros((per_rank,) + shape, dtype=torch.int32, device=buf_dev)
    for k, o in enumerate(outs):
        local[k] = o.cpu() if via_host else o
    full = torch.empty((world * per_rank,) + shape, dtype=torch.int32, device=buf_dev)
    dist.all_gather_into_tensor(full, local, group=group)
    if full.is_cuda:
        torch.cuda.synchronize(full.device)
    full = full.view((world, per_rank) + shape)
    # slot k of rank r is image r + k * world
    out = torch.stack([full[i % world, i // world] for i in range(n)])
    return out.to(dev), t_compute, time.perf_counter() - t1
