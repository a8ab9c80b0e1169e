#!/bin/bash
# Scaling sweep of bench.py on ONE node: N = 1, 2, 4, 8 (as many GPUs as the node has), weak (every rank its own 65,536 gates)
# and strong (one batch of 65,536 split), each as the driver launches it (torch.distributed.run, one rank per GPU, RCCL).
# Before the sweep: plain `python bench.py` against `--force-dist` at ONE rank (RCCL initialised, the output all-gather inside
# the timed region): the two values must agree within 2 %, or the N > 1 lines are not comparable with the N = 1 headline.
# usage: tools/scale_sweep.sh [out_dir] [steps]      (JSON lines land in out_dir/scale_<weak|strong>_<N>.json)
set -o pipefail
OUT="${1:-gpurun_out/scale_sweep}"; STEPS="${2:-5}"
mkdir -p "$OUT"
NGPU=$(python -c "import torch; print(torch.cuda.device_count())")
COMMON="--steps $STEPS --warmup 1 --cpu-sample 0 --no-mnist --no-live-traffic"
python bench.py $COMMON > "$OUT/plain_1.json" || exit 1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 $COMMON --force-dist > "$OUT/forced_dist_1.json" || exit 1
python - "$OUT" <<'PY' || exit 1
import json, sys
out = sys.argv[1]
a = json.loads(open(out + "/plain_1.json").read().strip().splitlines()[-1])
b = json.loads(open(out + "/forced_dist_1.json").read().strip().splitlines()[-1])
ratio = b["value"] / a["value"]
print("N=1 plain %.0f/s, under torch.distributed.run with RCCL gather %.0f/s, ratio %.4f; gather alone %.3f ms; gathered batch ok: %s"
      % (a["value"], b["value"], ratio, b["collective"]["ms_alone_unoverlapped"], b["checks"]["gathered_batch_ok"]))
assert abs(ratio - 1.0) <= 0.02 and b["checks"]["gathered_batch_ok"], "N = 1 under the launcher differs from the plain bench by more than 2 %"
PY
for N in 1 2 4 8; do
  [ "$N" -gt "$NGPU" ] && break
  for MODE in weak strong; do
    [ "$N" = 1 ] && [ "$MODE" = strong ] && continue
    # the weak line also carries BASELINE configs[4]: N encrypted CIFAR images, one per GPU, logits gathered (cifar_batch)
    CB="off"; [ "$MODE" = weak ] && CB="on"
    python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29550 + N)) bench.py --gpus $N $COMMON --scaling $MODE --cifar-batch $CB \
      > "$OUT/scale_${MODE}_$N.json" || exit 1
    # every line is checked before it is believed: the size that ran, the collective that ran, the gathered batch, and that no
    # rank's kernels were slower than the others' by 3 % (a slow rank sets the whole job's time)
    python - "$OUT/scale_${MODE}_$N.json" $N $MODE <<'PY' || exit 1
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
N, mode = int(sys.argv[2]), sys.argv[3]
br = [k["blind_rotate"] for k in d["kernels_ms_per_rank"]]
spread = (max(br) - min(br)) / max(br)
print("N=%d %-6s %10.0f bootstraps/s  %8.3f ms/step  kernel ms per rank: %s (spread %.1f %%)  gather alone %s ms"
      % (d["n_gpus"], d["scaling"], d["value"], d["ms_per_step"], br, 100 * spread, d["collective"] and d["collective"]["ms_alone_unoverlapped"]))
assert d["n_gpus"] == N and len(br) == N, "ran at another size than asked"
assert d["scaling"] == mode
if N > 1:
    assert d["collective"] and d["collective"]["backend"].startswith("nccl"), "the output gather did not run over RCCL"
    assert d["checks"]["gathered_batch_ok"], "the gathered batch is wrong"
    assert spread < 0.03, "per-rank kernel times differ by %.1f %%" % (100 * spread)
assert d["checks"]["all_outputs_decrypt_to_nand"]
cb = d.get("cifar_batch")
if mode == "weak":
    assert cb and cb["images"] == N and cb["logits_equal_single_gpu"], "the image-parallel CIFAR batch did not run or its gathered logits differ from single-GPU runs"
    print("      cifar_batch: %d image(s) in %.3f s = %.3f images/s, gather %.3f ms, per-rank compute ms %s"
          % (cb["images"], cb["s_per_batch"], cb["images_per_s"], cb["gather_ms"], [r["compute_ms"] for r in cb["per_rank_ms"]]))
PY
  done
done
