set -o pipefail
run() { name=$1; shift; timeout -k 10 300 python3 bench.py --no-cpu-b128 "$@" > gpurun_out/m_$name.log 2>&1; tail -1 gpurun_out/m_$name.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$name', round(d['value'],1), round(d['ms_per_step'],3), round(d['median_ms_per_step'],3))" ; }
run default
run library --bert-gemm library
run split2 --bert-gemm split2
run bf16 --dtype bf16
run gan --epoch 11
run v42 --dataset TED_expressive --batch 64
run v42gan --dataset TED_expressive --batch 64 --epoch 11
run v42ganbf16 --dataset TED_expressive --batch 64 --epoch 11 --dtype bf16
run feed --feed-host
run eager --eager
