"""Host CPU budget.  Containers routinely expose every core of the node (os.cpu_count() = 256 on the MI355X boxes)
under a much smaller CFS quota (16 CPUs there).  torch then sizes its intra-op OpenMP pool to 256 threads; their
spin-waiting burns the quota in a few milliseconds and the kernel freezes the WHOLE process — Python launch thread
included — until the next 100 ms period: measured as random 70-90 ms holes in the middle of training / rendering
steps whose GPU work is 5-40 ms.  The host side of this package is launch plumbing and needs a handful of threads."""
import os


def effective_cpus():
    """CPUs this process may actually use: min(affinity, cgroup quota)."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    quota = None
    try:                                            # cgroup v2
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(p)
    except (OSError, ValueError):
        try:                                        # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / p
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = min(n, max(1, int(quota)))
    return max(1, n)


def limit_host_threads():
    """Cap torch's intra-op pool to the CPU budget (NF_HOST_THREADS overrides); returns the thread count in effect."""
    import torch
    want = os.environ.get("NF_HOST_THREADS")
    n = int(want) if want else min(effective_cpus(), 16)
    local = int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1)       # one process per GPU shares the same budget
    if not want and local > 1:
        n = max(1, n // local)
    if torch.get_num_threads() > n:
        torch.set_num_threads(max(1, n))
    return torch.get_num_threads()
