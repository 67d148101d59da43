"""Achieved HBM bandwidth per kernel of the training step: PMC bytes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/train_prof.py,
collected by tools/train_pmc.sh; 2 x FETCH_SIZE + WRITE_SIZE, KiB -> bytes, as MI355X_MICROARCH.md prescribes) over the kernel's average
duration in the kernel trace of the same script (tools/profile_round.sh: gpurun_out/round/train_trace).
usage: python tools/train_traffic.py [gpurun_out/train_pmc] [gpurun_out/round/train_trace] > profiles/rNN_train_traffic.txt"""
import collections
import csv
import os
import re
import sys

base = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/train_pmc"
trace = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/round/train_trace"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 6          # steps in the kernel trace (train_prof.py: 1 warm-up + N timed [+ 1 marked])
pmc_steps = int(sys.argv[4]) if len(sys.argv) > 4 else 3      # steps in each PMC pass


def short(n):
    m = re.match(r"(?:void )?(?:\(anonymous namespace\)::|fear::)?([A-Za-z0-9_]+)(<[^>]*>)?", n)
    return (m.group(1) + (m.group(2) or "")) if m else n


def load(c):
    d = collections.defaultdict(lambda: [0, 0.0])
    sub = f"{base}/train_pmc_{c}" if os.path.isdir(f"{base}/train_pmc_{c}") else f"{base}/pmc_{c}"
    for r in csv.DictReader(open(f"{sub}/p_counter_collection.csv")):
        k = short(r["Kernel_Name"])
        d[k][0] += 1
        d[k][1] += float(r["Counter_Value"]) * 1024.0
    return d


F, W = load("FETCH_SIZE"), load("WRITE_SIZE")
dur = collections.defaultdict(list)
trace_rows = sorted(csv.DictReader(open(f"{trace}/p_kernel_trace.csv")), key=lambda r: int(r["Start_Timestamp"]))
# the steps begin with the stem's forward: what runs before it is the network's construction (one copy per parameter tensor into the
# flat buffer, the random initialisation) and does not belong to a step
first = next((i for i, r in enumerate(trace_rows) if "stem_im2col_kernel" in r["Kernel_Name"] or "pw_stat_kernel<1, true>" in r["Kernel_Name"]), 0)
for r in trace_rows[first:]:
    dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = []
for n, (c, f) in F.items():
    if n not in dur or n not in W:
        continue
    t = sum(dur[n]) / len(dur[n])
    byts = (2.0 * f + W[n][1]) / c
    rows.append((t * len(dur[n]) / steps, n, len(dur[n]) / steps, t, byts / 1e6, byts / t / 1e6))
rows.sort(reverse=True)
tot_bytes = sum((2.0 * f + W[n][1]) for n, (c, f) in F.items() if n in W) / pmc_steps
n_launch = sum(len(v) for v in dur.values()) / steps
t_kernels = sum(sum(v) for v in dur.values()) / steps
print(f"per step: {tot_bytes / 1e9:.1f} GB of PMC traffic (2 x FETCH_SIZE + WRITE_SIZE), {n_launch:.0f} launches, {t_kernels / 1e3:.2f} ms of kernel time "
      f"(summed over the streams) = {tot_bytes / t_kernels / 1e6:.2f} TB/s average")
print("ms/step  kernel                                        calls/step   avg us   MB/call   TB/s")
for ms, n, cs, t, mb, tb in rows[:40]:
    print(f"{ms / 1e3:6.2f}   {n[:44]:44s} {cs:9.0f} {t:8.1f} {mb:9.1f} {tb:6.2f}")
