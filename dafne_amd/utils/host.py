"""Host-side facts the engine's plumbing needs."""
import math
import os

import contextlib

__all__ = ["usable_cpus", "capped_torch_threads"]


def usable_cpus(cgroup_root="/sys/fs/cgroup"):
    """CPUs this process can actually keep busy: the scheduler affinity, capped by the cgroup CPU bandwidth quota
    (cgroup v2 `cpu.max`, v1 `cpu.cfs_quota_us / cpu.cfs_period_us`).  `os.cpu_count()` ignores both: the MI355X boxes of this
    project show 256 CPUs under a quota of 16, and every thread beyond the quota gets the WHOLE process group throttled --
    torch's 128-thread intra-op pool or 32 decode workers stall the HIP runtime's threads for tens of milliseconds."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    quota = None
    try:
        with open(os.path.join(cgroup_root, "cpu.max")) as f:
            q, p = f.read().split()[:2]
            if q != "max":
                quota = float(q) / float(p)
    except (OSError, ValueError):
        try:
            with open(os.path.join(cgroup_root, "cpu", "cpu.cfs_quota_us")) as f:
                q = float(f.read())
            with open(os.path.join(cgroup_root, "cpu", "cpu.cfs_period_us")) as f:
                p = float(f.read())
            if q > 0 and p > 0:
                quota = q / p
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = min(n, max(1, int(math.floor(quota))))
    return max(1, n)


@contextlib.contextmanager
def capped_torch_threads(cap=8):
    """torch's intra-op pool limited to min(cap, usable_cpus()) threads for the duration of the block, restored afterwards.
    The streamed evaluation loop's host side is latency work (pinned staging copies, a few small tensor ops per batch): with
    one pool worker per visible CPU the workers' spin-wait after every copy starves the HIP runtime's completion threads
    (50-200 ms stalls in the launch calls every few batches; profiles/NOTES_r04.md).  The cap used to be set process-wide and
    never restored from inside the detector (advisor, round 4); the LOOP owns it now (evaluation.inference.inference_on_dataset)."""
    import torch
    before = torch.get_num_threads()
    want = max(1, min(int(cap), usable_cpus()))
    if before > want:
        torch.set_num_threads(want)
    try:
        yield
    finally:
        if torch.get_num_threads() != before:
            torch.set_num_threads(before)
