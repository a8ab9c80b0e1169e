#!/bin/bash
# Round-end measurement set on ONE MI355X box -> gpurun_out/final/ (copy what is judged into profiles/).
# usage: tools/final_profile.sh PREFIX      (e.g. h)
set -o pipefail
P="${1:-h}"; OUT=gpurun_out/final; mkdir -p $OUT
export TMPDIR=/tmp
python bench.py > $OUT/${P}_wg_fft_bench.json 2> $OUT/bench_default.err && tail -c 400 $OUT/${P}_wg_fft_bench.json && echo
python bench.py --no-live-traffic --params redsec_small_v2 > $OUT/${P}_wg_fft_bench_redsec_params.json 2>> $OUT/bench_default.err
python bench.py --no-live-traffic --mode exact --cpu-sample 0 > $OUT/${P}_exact_ntt_bench.json 2>> $OUT/bench_default.err
python bench.py --no-live-traffic --mode exact --cpu-sample 0 --params redsec_small_v2 > $OUT/${P}_exact_ntt_bench_redsec_params.json 2>> $OUT/bench_default.err
echo "benches done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-exact-check --no-cifar > $OUT/prof_bench.log 2>&1
find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/${P}_wg_fft_kernel_stats.csv \;
head -4 $OUT/${P}_wg_fft_kernel_stats.csv
python tools/mnist_latency.py > $OUT/${P}_mnist_latency.txt 2>&1; head -1 $OUT/${P}_mnist_latency.txt
REDSEC_MODE=split python tools/mnist_latency.py > $OUT/${P}_mnist_latency_split_mode.txt 2>&1; head -1 $OUT/${P}_mnist_latency_split_mode.txt
python tools/mnist_cpu_baseline.py > $OUT/${P}_mnist_cpu_baseline.txt 2>&1; tail -2 $OUT/${P}_mnist_cpu_baseline.txt
python tools/cifar_profile.py binarynet $OUT/cifar_prof > $OUT/${P}_cifar_binarynet_driver.txt 2>&1; grep -E "wall|Classification" $OUT/${P}_cifar_binarynet_driver.txt
find $OUT/cifar_prof -name "*kernel_stats.csv" -exec cp {} $OUT/${P}_cifar_binarynet_driver_kernel_stats.csv \;
rm -rf $OUT/prof $OUT/cifar_prof
echo "final profile set done"
# round 3 additions: general ring kernels at full size (synthetic keys generated on the device), phase stamps if the variant is there
python tools/general_rate.py > $OUT/${P}_general_path_rates.jsonl 2> $OUT/general_rate.err; cat $OUT/${P}_general_path_rates.jsonl | cut -c1-220
if [ -f variants/lib_stamps.so ]; then
  REDSEC_HIP_LIB=$PWD/variants/lib_stamps.so python tools/stamp_profile.py default128 16384 > $OUT/${P}_stamps_wg_default128.json 2>/dev/null
  REDSEC_HIP_LIB=$PWD/variants/lib_stamps.so python tools/stamp_profile.py redsec_small_v2 16384 > $OUT/${P}_stamps_wg_redsec.json 2>/dev/null
fi
echo "round-3 additions done"
# round 6 additions: PCIe-inclusive rate of the host-pointer calls, the latency forms on both shipped sets
python tools/host_api_rate.py > $OUT/${P}_host_api_rate.txt 2>&1; tail -2 $OUT/${P}_host_api_rate.txt
python tools/small_batch_probe.py 196 1024 --reps 9 > $OUT/${P}_small_batches_redsec_set.txt 2>/dev/null; cat $OUT/${P}_small_batches_redsec_set.txt
python tools/small_batch_probe.py 196 256 --reps 9 --params default128 > $OUT/${P}_small_batches_default128.txt 2>/dev/null; cat $OUT/${P}_small_batches_default128.txt
echo "round-6 additions done"
