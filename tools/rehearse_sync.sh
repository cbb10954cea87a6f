# The N > 1 plan of the recorded step on a 1-rank RCCL group (bench.py --rehearse-sync: row-sharded mapping layer with all-gather of S,
# flat gradient copies + all-reduces between graph launches; default: the generator's backward cut at the decoder input with the first half's
# all-reduce started under the second half, --flat-exchange: one all-reduce behind the whole backward) next to the plain N = 1 step, same box,
# for the five workloads.
set -o pipefail
run() { name=$1; shift; timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 40 --warmup 8 "$@" > gpurun_out/rs_$name.log 2>&1; tail -1 gpurun_out/rs_$name.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$name', round(d['value'],1), round(d['ms_per_step'],3))" ; }
for cfg in "fp32:" "bf16:--dtype bf16" "gan:--epoch 11" "v42:--dataset TED_expressive --batch 64" "v42ganbf16:--dataset TED_expressive --batch 64 --epoch 11 --dtype bf16"; do
  n=${cfg%%:*}; a=${cfg#*:}
  run ${n}_plain $a
  run ${n}_sync $a --rehearse-sync
  run ${n}_sync_flat $a --rehearse-sync --flat-exchange
done
