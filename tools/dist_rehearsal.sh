#!/bin/bash
# One-GPU rehearsal of the N > 1 bench paths (the driver launches the real ones on an 8-GPU node):
#   (1) N = 1 under torch.distributed.run with RCCL initialised and the output all-gather inside the timed region (--force-dist);
#   (2) two ranks sharing device 0 over gloo (REDSEC_BENCH_REHEARSAL=1), weak and strong: the sharding, padding and gather checks;
#   both carry the cifar_batch leg (BASELINE configs[4]: one encrypted CIFAR image per rank, logits gathered, re-checked against single runs).
# Timings of (2) say nothing about scaling.
set -o pipefail
OUT="${1:-gpurun_out/rehearsal}"; mkdir -p "$OUT"
COMMON="--steps 2 --warmup 1 --cpu-sample 0 --no-mnist --no-live-traffic --no-exact-check"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 $COMMON --force-dist --cifar-batch on > "$OUT/rccl_world1.json" 2> "$OUT/rccl_world1.err" || { tail -5 "$OUT/rccl_world1.err"; exit 1; }
for MODE in weak strong; do
  REDSEC_BENCH_REHEARSAL=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29542 bench.py --gpus 2 $COMMON --scaling $MODE --gates 16384 --cifar-batch-net binarynet_small \
    > "$OUT/gloo_2ranks_$MODE.json" 2> "$OUT/gloo_2ranks_$MODE.err" || { tail -5 "$OUT/gloo_2ranks_$MODE.err"; exit 1; }
done
python - "$OUT" <<'PY'
import json, sys
out = sys.argv[1]
for name in ("rccl_world1", "gloo_2ranks_weak", "gloo_2ranks_strong"):
    d = json.loads(open("%s/%s.json" % (out, name)).read().strip().splitlines()[-1])
    print(name, "n_gpus", d["n_gpus"], d["scaling"], round(d["value"]), "bootstraps/s; backend", d["collective"]["backend"], "gathered ok:", d["checks"]["gathered_batch_ok"],
          "decrypt ok:", d["checks"]["all_outputs_decrypt_to_nand"], "per-rank kernel ms", [k["blind_rotate"] for k in d["kernels_ms_per_rank"]])
    assert d["checks"]["gathered_batch_ok"] and d["checks"]["all_outputs_decrypt_to_nand"]
    cb = d["cifar_batch"]
    print("   cifar_batch:", cb["images"], "image(s),", cb["s_per_batch"], "s,", cb["collective"], "gather", cb["gather_ms"], "ms; equal to single-GPU runs:", cb["logits_equal_single_gpu"])
    assert cb["logits_equal_single_gpu"] and cb["images"] == d["n_gpus"]
PY
