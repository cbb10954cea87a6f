#!/bin/bash
# bf16 configurations with the weight gradients taken straight from the GEMM's fp32 accumulators (HOPMI_MM_F32=1: no bf16 rounding of
# dW, no cast-back launch): add those GEMMs' shapes to a copy of the shipped table, then A/B the bf16 step with and without them.
set -e
mkdir -p gpurun_out
cp hop-*/tuned/gemm_gfx950.csv gpurun_out/tunableop_f32out0.csv
export HOPMI_MM_F32=1
export PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME=gpurun_out/tunableop_f32out.csv
for cfg in "--dataset TED --batch 128 --epoch 0 --dtype bf16" "--dataset TED --batch 128 --epoch 11 --dtype bf16" \
           "--dataset TED_expressive --batch 64 --epoch 11 --dtype bf16"; do
  echo "== $cfg"
  timeout -k 10 600 python3 bench.py --eager --kernel-steps 0 --steps 2 --warmup 1 --no-cpu-baseline $cfg 2>/dev/null | tail -1 | cut -c1-120
  wc -l gpurun_out/tunableop_f32out0.csv
done
unset PYTORCH_TUNABLEOP_ENABLED PYTORCH_TUNABLEOP_TUNING PYTORCH_TUNABLEOP_FILENAME
for v in 0 1 0 1; do
  HOPMI_MM_F32=$v python3 bench.py --no-cpu-baseline --steps 40 --warmup 8 --dtype bf16 --tuned-table gpurun_out/tunableop_f32out0.csv 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('HOPMI_MM_F32=$v bf16', round(d['value'],1), round(d['ms_per_step'],3), d['losses'])"
done
for v in 0 1; do
  HOPMI_MM_F32=$v python3 bench.py --no-cpu-baseline --steps 40 --warmup 8 --dtype bf16 --dataset TED_expressive --batch 64 --epoch 11 --tuned-table gpurun_out/tunableop_f32out0.csv 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('HOPMI_MM_F32=$v v42 gan bf16', round(d['value'],1), round(d['ms_per_step'],3), d['losses'])"
done
