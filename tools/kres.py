"""Kernel resource table of the library build: python tools/kres.py [name-substring ...]
(hipcc -Rpass-analysis=kernel-resource-usage of csrc/fear_engine.hip: VGPRs, spills, scratch, occupancy, LDS per kernel)."""
import re
import subprocess
import sys

cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value", "-fno-honor-nans",
       "-Rpass-analysis=kernel-resource-usage", "-o", "/tmp/kres.so", "feartracker_amd/csrc/fear_engine.hip"]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in err.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
if "error" in err:
    print("\n".join(l for l in err.splitlines() if "error" in l)[:4000])
want = sys.argv[1:]
dem = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
for r, d in zip(rows, dem):
    if want and not any(w in d for w in want):
        continue
    d = re.sub(r"\(anonymous namespace\)::|fear::|void ", "", d)
    d = re.sub(r"\(.*", "", d)
    print(f"{d[:70]:70s} vgpr {r.get('VGPRs', 0):4d} agpr {r.get('AGPRs', 0):4d} spill {r.get('VGPRs Spill', 0):5d} scratch {r.get('ScratchSize', 0):5d} occ {r.get('Occupancy', 0):2d} lds {r.get('LDS Size', 0):6d}")
