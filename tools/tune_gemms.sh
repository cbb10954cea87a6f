#!/bin/bash
# Regenerate the GEMM selection table on an MI355X (through gpurun, from the repository root):
#   bash tools/tune_gemms.sh   ->  gpurun_out/tunableop_results0.csv  (copy to <package>/tuned/gemm_gfx950.csv)
# Every workload of BASELINE.json runs 2 steps with TunableOp tuning on; results accumulate in one file.
set -e
export PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME=gpurun_out/tunableop_results.csv
mkdir -p gpurun_out
rm -f gpurun_out/tunableop_results*.csv
for cfg in "--dataset TED --batch 128 --epoch 0" "--dataset TED --batch 128 --epoch 11" \
           "--dataset TED_expressive --batch 64 --epoch 0" "--dataset TED_expressive --batch 64 --epoch 11" \
           "--dataset TED --batch 128 --epoch 0 --dtype bf16" "--dataset TED --batch 128 --epoch 11 --dtype bf16" \
           "--dataset TED_expressive --batch 64 --epoch 11 --dtype bf16"; do
  echo "== tuning: $cfg"
  timeout -k 10 600 python bench.py --eager --kernel-steps 0 --steps 2 --warmup 1 --no-cpu-baseline $cfg 2>/dev/null | tail -1 | cut -c1-160
  wc -l gpurun_out/tunableop_results0.csv
done
