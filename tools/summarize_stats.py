#!/usr/bin/env python3
"""One rocprofv3 --kernel-trace --stats run (raw CSVs under <dir>) -> profiles/<tag>_<name>_kernel_stats.csv (rows >= 0.05 % of the
kernel time, names truncated).  usage: summarize_stats.py <tag> <name> <dir> <steps profiled> "<command line / workload>" """
import csv, glob, os, sys

csv.field_size_limit(1 << 30)
tag, name, root, nsteps, what = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5]
hits = glob.glob(os.path.join(root, "**", "*kernel_stats.csv"), recursive=True)
if not hits:
    raise SystemExit(f"no kernel_stats.csv under {root}")
rows = list(csv.DictReader(open(sorted(hits, key=os.path.getsize)[-1])))
total = sum(float(r["TotalDurationNs"]) for r in rows)
nsteps = sum(int(r["Calls"]) for r in rows if "hop_losses_fwd_kernel" in r["Name"]) or nsteps     # one launch per train_llm step
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", f"{tag}_{name}_kernel_stats.csv")
with open(out, "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats --output-format csv -- {what}\n")
    f.write(f"# {nsteps} train_llm steps profiled (eager steps of the kernel region and the warm-up + replays of the recorded step); "
            f"total kernel time {total / 1e6:.2f} ms = {total / 1e6 / nsteps:.2f} ms per step\n")
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
        if float(r["TotalDurationNs"]) >= 5e-4 * total:
            w.writerow([r["Name"][:120], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
print(out, f"{total / 1e6 / nsteps:.2f} ms per step")
