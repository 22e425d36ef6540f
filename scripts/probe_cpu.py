import os, time, torch
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads(), flush=True)
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(p, open(p).read().strip(), flush=True)
    except Exception as e: print(p, "n/a", flush=True)
a = torch.randn(2048, 2048); b = torch.randn(2048, 2048)
for n in (8, 16, 32, 64):
    torch.set_num_threads(n)
    a @ b
    t = time.perf_counter()
    for _ in range(5): a @ b
    dt = (time.perf_counter() - t) / 5
    print(f"threads {n}: {2*2048**3/dt/1e9:.1f} GFLOP/s", flush=True)
