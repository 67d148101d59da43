#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected separately,
MI355X_MICROARCH.md §HBM / §rocprofv3 PMC slots).

Units and gfx950 correction as that guide prescribes: the counters are in KiB; on gfx950 FETCH_SIZE reports
exactly half of the bytes of wide (16 B/lane) coalesced streaming reads, so the read side is doubled;
WRITE_SIZE is taken as is (uncalibrated per the guide).  Output: JSON {kernel symbol: {fetch_kib_raw,
write_kib_raw, traffic_bytes_per_launch, launches}} for the fear:: kernels, averaged over launches.

usage: pmc_to_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> > traffic.json
"""
import csv
import json
import re
import sys
from collections import defaultdict


def per_kernel(path, counter):
    acc = defaultdict(lambda: [0.0, 0])
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"]
            if "fear::" not in name:
                continue
            name = re.sub(r"\(.*", "", name).replace("void ", "")
            a = acc[name]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(k, (0.0, 0))
        w, nw = write.get(k, (0.0, 0))
        out[k] = {"fetch_kib_raw": f, "write_kib_raw": w, "launches": max(nf, nw),
                  "traffic_bytes_per_launch": (2.0 * f + w) * 1024.0,
                  "correction": "2 x FETCH_SIZE (gfx950 wide-read undercount) + WRITE_SIZE, KiB -> bytes"}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
