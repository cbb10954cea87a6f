#!/usr/bin/env python3
"""A/B of environment settings on one box: tools/ab_env.py "NAME=a,b,c" [bench args...] -> ms per step for each value."""
import json, os, subprocess, sys
name, values = sys.argv[1].split("=")
for v in values.split(","):
    env = dict(os.environ, **{name: v})
    r = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline"] + sys.argv[2:], env=env, capture_output=True, text=True)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
        rf = d.get("roofline") or {}
        print(f"{name}={v}: {d['value']:.0f} clips/s {d['ms_per_step']:.3f} ms/step  wn_stack {rf.get('avg_us', 0):.1f} us  wn_bwd {rf.get('wn_layer_bwd_avg_us', 0):.1f}"
              f" attn f/b {rf.get('reprog_attn_fwd_avg_us', 0):.0f}/{rf.get('reprog_attn_bwd_avg_us', 0):.0f} gru f/b {rf.get('gru_fwd_avg_us', 0):.0f}/{rf.get('gru_bwd_avg_us', 0):.0f}", flush=True)
    except Exception as e:          # noqa: BLE001
        print(f"{name}={v}: failed ({e}); stderr tail: {r.stderr[-400:]}", flush=True)
