#!/bin/bash
# Add the GEMM shapes the shipped table does not hold yet (through gpurun, from the repository root): TunableOp loads the
# shipped table, tunes only what it does not find while every BASELINE.json workload runs 2 eager steps, and writes the
# union to gpurun_out/tunableop_merged0.csv (copy over <package>/tuned/gemm_gfx950.csv).
set -e
mkdir -p gpurun_out
cp hop-*/tuned/gemm_gfx950.csv gpurun_out/tunableop_merged0.csv
export PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME=gpurun_out/tunableop_merged.csv
for cfg in "--dataset TED --batch 128 --epoch 0" "--dataset TED --batch 128 --epoch 11" \
           "--dataset TED_expressive --batch 64 --epoch 0" "--dataset TED_expressive --batch 64 --epoch 11" \
           "--dataset TED --batch 128 --epoch 0 --dtype bf16" "--dataset TED --batch 128 --epoch 11 --dtype bf16" \
           "--dataset TED_expressive --batch 64 --epoch 11 --dtype bf16"; do
  echo "== $cfg"
  timeout -k 10 600 python3 bench.py --eager --kernel-steps 0 --steps 2 --warmup 1 --no-cpu-baseline $cfg 2>/dev/null | tail -1 | cut -c1-120
  wc -l gpurun_out/tunableop_merged0.csv
done
