#!/usr/bin/env python3
"""One-crop `fear_track` timeline (the "tiny" launch plan): how profiles/rNN_batch1_timeline.txt is made.

  on the GPU box:   cd /tmp && rocprofv3 --kernel-trace -d out -o p --output-format csv -- python tools/b1_timeline.py run
  anywhere:         python tools/b1_timeline.py fold out/p_kernel_trace.csv > profiles/rNN_batch1_timeline.txt
`run` also prints the un-profiled time per call and the per-op HIP-event times of the plan.
"""
import csv
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run():
    import torch
    from feartracker_amd import DEFAULT_WEIGHTS, FEARNetHIP
    net = FEARNetHIP(DEFAULT_WEIGHTS, device=0, max_batch=1)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 3, 256, 256, generator=g).cuda()
    z = net.get_features(torch.randn(1, 3, 128, 128, generator=g).cuda())
    for _ in range(30):
        net.track_maps(x, z)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        net.track_maps(x, z)
    torch.cuda.synchronize()
    print(f"ms per one-crop track: {(time.perf_counter() - t0) / 200 * 1e3:.4f}", file=sys.stderr)
    net.set_plan_crops(1)
    plan = net.plan(256, True)
    net.set_profile(True)
    for _ in range(50):
        net.track_maps(x, z)
    torch.cuda.synchronize()
    for (name, _, _), (ms, n) in zip(plan, net.profile_read(256, True)):
        print(f"{name:40s} {ms / max(n, 1) * 1e3:7.1f} us (bracketed by events)", file=sys.stderr)


def fold(path):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    first = [i for i, r in enumerate(rows) if "ir_tile_v2_kernel<27" in r["Kernel_Name"]]
    seq = rows[first[-51]: first[-50]] if len(first) > 51 else rows[first[-1]:]     # a call from before the event-bracketed ones
    t0 = int(seq[0]["Start_Timestamp"])
    print("# rocprofv3 --kernel-trace of one batch-1 fear_track call (tiny plan, exact fp32), MI355X; start us | duration us | grid | queue | kernel")
    for r in seq:
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].replace("fear::", "").replace("void ", "")[:90]
        print(f"{(st - t0) / 1e3:8.1f}  {(en - st) / 1e3:6.1f}  {r['Grid_Size_X'] + 'x' + r['Grid_Size_Y']:9s} q{r['Queue_Id']} {name}")
    print(f"# span {(int(seq[-1]['End_Timestamp']) - t0) / 1e3:.1f} us, {len(seq)} kernels")


if __name__ == "__main__":
    if len(sys.argv) >= 2 and sys.argv[1] == "run":
        run()
    elif len(sys.argv) >= 3 and sys.argv[1] == "fold":
        fold(sys.argv[2])
    else:
        sys.exit(__doc__)
