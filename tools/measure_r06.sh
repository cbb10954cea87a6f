# Round-6 re-measurements on finite trajectories (one box): the graded kernel's batch sweep (exchange floor vs streaming rate) and
# the four size thresholds of the fp16 forms, each A/B-ed through bench.py (which now refuses to report a non-finite run).
set -o pipefail
mkdir -p gpurun_out
{
for B in 128 256 512 1024; do python3 tools/bench_wn_stack.py --V 9 --B $B 2>/dev/null | grep -E "persistent grid|one launch"; done
for B in 64 256; do python3 tools/bench_wn_stack.py --V 42 --B $B 2>/dev/null | grep -E "persistent grid|one launch"; done
} > gpurun_out/r06_wn_stack_sweep.txt 2>&1
cat gpurun_out/r06_wn_stack_sweep.txt
{
echo "# configs[1] (TED V=9 B=128 fp32), 40 steps after 8"
python3 tools/ab_env.py "HOPMI_IMG_MIN_ROWS=3072,100000,3072,100000" --steps 40 --warmup 8
python3 tools/ab_env.py "HOPMI_LINEAR_IMG_MIN_ROWS=2048,100000,2048,100000" --steps 40 --warmup 8
python3 tools/ab_env.py "HOPMI_F16_LINEAR_MIN_MNK=3.0e9,5.0e9,1.0e9,3.0e9" --steps 40 --warmup 8
python3 tools/ab_env.py "HOPMI_F16_TN_MIN_MNK=1.0e9,2.0e9,5.0e8,1.0e9" --steps 40 --warmup 8
python3 tools/ab_env.py "HOPMI_GEMM_AB_WAVES=8,4,8,4" --steps 40 --warmup 8
echo "# configs[3] (TED-Expressive V=42 B=64 fp32)"
python3 tools/ab_env.py "HOPMI_IMG_MIN_ROWS=3072,1024,3072,1024" --steps 40 --warmup 8 --dataset TED_expressive --batch 64
python3 tools/ab_env.py "HOPMI_LINEAR_IMG_MIN_ROWS=2048,100000,2048,100000" --steps 40 --warmup 8 --dataset TED_expressive --batch 64
} > gpurun_out/r06_thresholds.txt 2>&1
cat gpurun_out/r06_thresholds.txt
