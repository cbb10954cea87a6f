# One bench.py run per BASELINE.json configuration / A-B variant: name, clips/s, ms per step (mean, median), the last step's losses
# and bench.py's exit code (3 = the model left the timed region non-finite: no throughput is reported for such a run).
set -o pipefail
run() { name=$1; shift; timeout -k 10 300 python3 bench.py --no-cpu-b128 "$@" > gpurun_out/m_$name.log 2>&1; rc=$?; tail -1 gpurun_out/m_$name.log | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read())
except Exception as e:
    print('$name', 'NO JSON LINE', 'rc', $rc); sys.exit(0)
v=d.get('value')
print('$name', None if v is None else round(v,1), round(d.get('ms_per_step',0),3), round(d.get('median_ms_per_step',0),3), 'losses', {k:(round(x,3) if isinstance(x,float) else x) for k,x in d.get('losses',{}).items()}, 'rc', $rc, d.get('error',''))" ; }
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
