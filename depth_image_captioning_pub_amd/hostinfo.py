"""Host-side facts: how many CPU cores this process may actually use (cgroup quota aware)."""
from __future__ import annotations

import os


def host_cores() -> int:
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:                                   # cgroup v2: "<quota> <period>" or "max <period>"
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:                               # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return max(1, n)


def visible_gpus():
    """Number of GPUs this process would see, WITHOUT any HIP / HSA call (a parent that has touched the runtime must not fork
    GPU ranks): the device-visibility environment variables if one is set, else the KFD topology in sysfs (nodes with SIMDs
    are GPUs; no KFD topology at all = no AMD GPU driver = 0).  None only when the topology exists but cannot be read - the
    caller then lets the ranks themselves find out."""
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([t for t in v.split(",") if t.strip() != ""])
    root = "/sys/class/kfd/kfd/topology/nodes"
    if not os.path.isdir(root):
        return 0
    try:
        n = 0
        for node in os.listdir(root):
            for line in open(os.path.join(root, node, "properties")):
                key, _, val = line.partition(" ")
                if key == "simd_count" and int(val) > 0:
                    n += 1
        return n
    except Exception:
        return None
