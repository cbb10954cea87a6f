#!/bin/bash
# Runs tools/nan_hunt.py once per variant ("TAG VAR=VAL ... [-- extra args]"), each as its own child process, logs under gpurun_out/hunt/.
# Stops at the first variant that times out (a hung GPU step must not be followed by another one).
mkdir -p gpurun_out/hunt
while IFS= read -r line; do
  [ -z "$line" ] && continue
  tag=${line%% *}; rest=${line#* }; [ "$rest" = "$line" ] && rest=""
  envs=${rest%%--*}; extra=""; case "$rest" in *--*) extra=${rest#*--};; esac
  echo "=== $tag: $envs $extra"
  env $envs timeout -k 10 ${HUNT_TIMEOUT:-200} python3 tools/nan_hunt.py --tag "$tag" $extra > gpurun_out/hunt/$tag.log 2>&1
  rc=$?
  grep -h "^HUNT" gpurun_out/hunt/$tag.log || tail -3 gpurun_out/hunt/$tag.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "variant $tag timed out (rc $rc): stopping"; exit 1; fi
done
exit 0
