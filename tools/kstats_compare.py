"""Per-kernel time of a rocprofv3 --stats run next to a committed profile (both normalised per profiled step = launches of wn_stack_fwd)."""
import csv, glob, re, sys


def load(path):
    if path.endswith(".csv"):
        f = path
    else:
        f = sorted(glob.glob(path + "/**/*kernel_stats.csv", recursive=True))[0]
    rows = {}
    with open(f) as fh:
        lines = [ln for ln in fh if not ln.startswith("#")]
    for r in csv.DictReader(lines):
        name = re.sub(r"\(.*", "", r["Name"]).replace("void ", "").strip()[:70]
        rows[name] = (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3)
    return rows


def steps(rows):
    for k, (c, t) in rows.items():
        if "wn_stack_fwd_kernel" in k:
            return c
    return 16


new, old = load(sys.argv[1]), load(sys.argv[2])
sn, so = steps(new), steps(old)
names = sorted(set(new) | set(old), key=lambda k: -(new.get(k, (0, 0))[1] / sn + old.get(k, (0, 0))[1] / so))
tn = sum(t for _, t in new.values()) / sn
to = sum(t for _, t in old.values()) / so
print(f"{'kernel':70s} {'new us/step':>12s} {'calls/step':>10s} {'avg us':>8s} | {'old us/step':>12s} {'calls/step':>10s} {'avg us':>8s}")
print(f"{'TOTAL':70s} {tn:12.1f} {'':10s} {'':8s} | {to:12.1f}")
for k in names[:70]:
    cn, tnn = new.get(k, (0, 0.0))
    co, too = old.get(k, (0, 0.0))
    print(f"{k:70s} {tnn / sn:12.1f} {cn / sn:10.1f} {tnn / cn if cn else 0:8.1f} | {too / so:12.1f} {co / so:10.1f} {too / co if co else 0:8.1f}")
