"""How many host CPUs does this process really have?  The GPU boxes of the pool show 256 logical CPUs (os.cpu_count(), the
scheduler affinity mask) behind a cgroup quota of 16: a torch CPU op that starts 128 intra-op threads there runs up to 30 x slower
than with 16 (57 s against 1.9 s for one fp32 oracle forward of the test-size UNet, measured in round 6).  Everything in this repo that
computes on the HOST - the oracle in tests/, bench.py's cpu_baseline leg - sizes its thread pool with host_cpus()."""
import os


def _cgroup_quota():
    try:                                             # cgroup v2
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()[:2]
        if q != "max":
            return max(1, int(int(q) / int(p)))
    except (OSError, ValueError):
        pass
    try:                                             # cgroup v1
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = int(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            p = int(f.read())
        if q > 0 and p > 0:
            return max(1, q // p)
    except (OSError, ValueError):
        pass
    return None


def host_cpus() -> int:
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    q = _cgroup_quota()
    return max(1, min(n, q) if q else n)


def cpu_facts() -> dict:
    return {"os_cpu_count": os.cpu_count(), "cgroup_cpu_quota": _cgroup_quota(), "host_cpus": host_cpus()}
