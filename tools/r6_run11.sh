#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6l
rm -rf "$O"; mkdir -p "$O"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_abi.py -m gpu -x -q > "$O/gputests.txt" 2>&1
echo "pytest rc $?" >> "$O/gputests.txt"
grep -n "passed\|failed\|FAILED\|rc \|Error" "$O/gputests.txt" | head
python - <<'PY'
import time, torch, sys
sys.path.insert(0, ".")
from feartracker_amd import DEFAULT_WEIGHTS, FEARNetHIP
g = torch.Generator().manual_seed(0)
x = torch.randn(1, 3, 256, 256, generator=g).cuda()
for rep in range(2):
    for on in (True, False):
        net = FEARNetHIP(DEFAULT_WEIGHTS, device=0, max_batch=1)
        net.set_fuse_reduce(on)
        z = net.get_features(torch.randn(1, 3, 128, 128, generator=g).cuda())
        for _ in range(50): net.track_maps(x, z)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(500): net.track_maps(x, z)
        torch.cuda.synchronize()
        print(f"fuse_reduce={int(on)}: {(time.perf_counter() - t0) / 500 * 1e3:.4f} ms per one-crop track call")
PY
