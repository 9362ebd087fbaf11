"""utils/host.py: the CPU count the engine's host-side pools are sized by (scheduler affinity capped by the cgroup quota)."""
import os

from dafne_amd.utils.host import usable_cpus


def _affinity():
    try:
        return len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return os.cpu_count() or 1


def test_usable_cpus_honours_the_cgroup_v2_quota(tmp_path):
    n = _affinity()
    assert usable_cpus(str(tmp_path)) == n                      # no cgroup files: the affinity
    (tmp_path / "cpu.max").write_text("max 100000\n")
    assert usable_cpus(str(tmp_path)) == n                      # unlimited
    (tmp_path / "cpu.max").write_text("150000 100000\n")         # 1.5 CPUs -> 1
    assert usable_cpus(str(tmp_path)) == 1
    (tmp_path / "cpu.max").write_text("1600000 100000\n")        # the MI355X boxes of this project: 16 of 256
    assert usable_cpus(str(tmp_path)) == min(n, 16)
    (tmp_path / "cpu.max").write_text("garbage\n")
    assert usable_cpus(str(tmp_path)) == n


def test_usable_cpus_honours_the_cgroup_v1_quota(tmp_path):
    n = _affinity()
    (tmp_path / "cpu").mkdir()
    (tmp_path / "cpu" / "cpu.cfs_quota_us").write_text("-1\n")
    (tmp_path / "cpu" / "cpu.cfs_period_us").write_text("100000\n")
    assert usable_cpus(str(tmp_path)) == n
    (tmp_path / "cpu" / "cpu.cfs_quota_us").write_text("300000\n")
    assert usable_cpus(str(tmp_path)) == min(n, 3)


def test_usable_cpus_on_this_host_is_sane():
    assert 1 <= usable_cpus() <= (os.cpu_count() or 1)
