"""Per-kernel table of a rocprofv3 --kernel-trace --stats run of tools/train_prof.py: python tools/kstats.py <dir> [steps_in_trace]"""
import csv, re, sys
d = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
rows = list(csv.DictReader(open(f"{d}/p_kernel_stats.csv")))
def short(n):
    m = re.match(r"(?:void )?(?:\(anonymous namespace\)::|fear::)?([A-Za-z0-9_]+)(<[^>]*>)?", n)
    return (m.group(1) + (m.group(2) or "")) if m else n
tot = sum(int(r['TotalDurationNs']) for r in rows)
n = sum(int(r['Calls']) for r in rows)
print(f"total {tot / steps / 1e6:.2f} ms of kernels per step, {n / steps:.0f} launches per step")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 32]:
    print(f"{int(r['TotalDurationNs']) / steps / 1e6:7.3f} ms/step {int(r['Calls']) / steps:7.1f} calls  avg {float(r['AverageNs']) / 1e3:8.1f} us  max {int(r['MaxNs']) / 1e3:8.1f}  {short(r['Name'])}")
