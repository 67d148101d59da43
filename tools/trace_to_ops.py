#!/usr/bin/env python3
"""Fold a rocprofv3 --kernel-trace CSV into a per-launch-plan-op summary.

rocprofv3's own --stats groups by kernel *symbol*; several layers share one template instantiation
(e.g. every 256x256 pointwise of the head runs `pw_mfma_kernel<2,8,false>`), so this tool walks the
dispatches in order, keeps the `fear::` kernels of the bench's timed steps (the last steps*plan_len
dispatches of the track plan) and averages by position inside the plan.  Op names come from the
`--dump-ops` table bench.py prints to stderr.

usage: trace_to_ops.py <kernel_trace.csv> <bench_stderr.log> <steps> [tail_iters] > per_op.csv
(tail_iters = plan executions AFTER the timed loop to skip: bench.py re-runs the steps in the other arithmetic
mode afterwards, max(3, warmup//2) warm-ups + steps)
"""
import csv
import re
import sys


def main():
    trace, log, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
    tail = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    names = []
    for line in open(log):
        m = re.match(r"\s*(\d+)\s+(\S+)\s+([\d.]+) ms/step", line)
        if m:
            names.append(m.group(2))
    plan_len = len(names)
    rows = []
    with open(trace) as fh:
        for r in csv.DictReader(fh):
            if r["Kernel_Name"].startswith(("void fear::", "fear::")):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                             int(r["Grid_Size_X"]), int(r["VGPR_Count"])))
    rows.sort()
    if tail:
        rows = rows[:-tail * plan_len]
    rows = rows[-steps * plan_len:]
    assert len(rows) == steps * plan_len, (len(rows), steps, plan_len)
    w = csv.writer(sys.stdout)
    w.writerow(["op", "name", "kernel", "grid_x", "vgpr", "launches", "avg_ns", "min_ns", "max_ns", "share_pct"])
    per = []
    for i in range(plan_len):
        d = [rows[s * plan_len + i][1] - rows[s * plan_len + i][0] for s in range(steps)]
        per.append((sum(d) / len(d), min(d), max(d)))
    total = sum(p[0] for p in per)
    for i in range(plan_len):
        kern = re.sub(r"\(.*", "", rows[i][2]).replace("void ", "")
        w.writerow([i, names[i], kern, rows[i][3], rows[i][4], steps, f"{per[i][0]:.0f}", per[i][1], per[i][2],
                    f"{100 * per[i][0] / total:.2f}"])
    w.writerow(["", "TOTAL", "", "", "", steps, f"{total:.0f}", "", "", "100.00"])


if __name__ == "__main__":
    main()
