#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repository root):
#   1. kernel-trace + stats of the default bench workload (kernel region: 7 eager steps; warm-up + timed region: graph replays)
#   2./3. PMC passes (FETCH_SIZE, WRITE_SIZE: separate passes, no other trace domains) for HBM traffic per launch,
#         on eager steps (--eager --kernel-steps 0: every dispatch is an ordinary launch)
#   4. PMC pass SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE for the matrix-pipe utilisation of every kernel family
#   5./6./7. kernel-trace + stats of the bf16 (configs[2] per GPU), TED-Expressive (configs[3]) and GAN-phase workloads
# Raw output goes to gpurun_out/prof_*; tools/summarize_profiles.py turns it into profiles/<tag>_*.{csv,json}.
set -e
TAG=${1:-r06}
export TMPDIR=/tmp
REPO=$(pwd)
mkdir -p gpurun_out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_stats -o stats -- python3 $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $REPO/gpurun_out/prof_stats.log 2>&1
echo "stats pass done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $REPO/gpurun_out/prof_fetch -o fetch -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --eager --kernel-steps 0 > $REPO/gpurun_out/prof_fetch.log 2>&1
echo "fetch pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $REPO/gpurun_out/prof_write -o write -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --eager --kernel-steps 0 > $REPO/gpurun_out/prof_write.log 2>&1
echo "write pass done"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $REPO/gpurun_out/prof_mfma -o mfma -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --eager --kernel-steps 0 > $REPO/gpurun_out/prof_mfma.log 2>&1
echo "mfma pass done"
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_bf16 -o stats -- python3 $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --dtype bf16 > $REPO/gpurun_out/prof_bf16.log 2>&1
echo "bf16 stats pass done"
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_v42 -o stats -- python3 $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --dataset TED_expressive --batch 64 > $REPO/gpurun_out/prof_v42.log 2>&1
echo "v42 stats pass done"
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_gan -o stats -- python3 $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --epoch 11 > $REPO/gpurun_out/prof_gan.log 2>&1
echo "gan stats pass done"
cd $REPO
python3 tools/summarize_profiles.py $TAG gpurun_out
python3 tools/summarize_stats.py $TAG bf16 gpurun_out/prof_bf16 16 "python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --dtype bf16  (configs[2] per GPU: TED V=9, B=128, bf16; 1x MI355X)"
python3 tools/summarize_stats.py $TAG v42 gpurun_out/prof_v42 16 "python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --dataset TED_expressive --batch 64  (configs[3]: V=42, B=64, fp32; 1x MI355X)"
python3 tools/summarize_stats.py $TAG gan gpurun_out/prof_gan 16 "python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --epoch 11  (configs[1] in the GAN phase: TED V=9, B=128, fp32, epoch 11; 1x MI355X)"
