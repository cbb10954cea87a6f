# A/B of one environment switch on one box: tools/ab_env.sh VAR "bench args" -> clips/s and ms/step for VAR=1,0,1,0
VAR=$1; shift
for v in 1 0 1 0; do
  env $VAR=$v python3 bench.py --no-cpu-baseline --steps 40 --warmup 8 "$@" 2>/dev/null | tail -1 > gpurun_out/ab_tmp.json
  python3 -c "
import json
d=json.loads(open('gpurun_out/ab_tmp.json').read()); print('$VAR=$v', round(d['value'],1), round(d['ms_per_step'],3))"
done
