#!/bin/bash
# what the GPU box's container is allowed to use of the host CPUs (bench.py sizes the node-level cpu_baseline from it)
echo "nproc: $(nproc)  getconf: $(getconf _NPROCESSORS_ONLN)"
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
echo "cfs quota/period: $(cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null) / $(cat /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null)"
echo "cpuset: $(cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null)"
python - <<'PY'
import os
print("affinity:", len(os.sched_getaffinity(0)), "cpu_count:", os.cpu_count())
print("loadavg:", os.getloadavg())
PY
lscpu | grep -E "Model name|Socket|Core|Thread|NUMA node\(s\)"
free -g | head -2
