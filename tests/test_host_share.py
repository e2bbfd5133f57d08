"""Host side of a rank (Weak #6): the native engine sizes its polling by the CPUs THIS rank can
count on -- min(affinity mask, cgroup CPU quota / LOCAL_WORLD_SIZE) -- not by the affinity mask alone (a GPU
box shows 256 CPUs to a container that owns 16).  No GPU: pconv_ee_host_cpus / pconv_ee_spin_us only read the
environment (include/pconv_hip.h)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def lib():
    from pseudocylindrical_convolution_amd import _native
    return _native.hip_lib()


@pytest.fixture()
def quota_file(tmp_path, monkeypatch):
    def write(text, local_world=None):
        p = tmp_path / "cpu.max"
        p.write_text(text)
        monkeypatch.setenv("PCONV_CGROUP_CPU_MAX", str(p))
        if local_world is None:
            monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
        else:
            monkeypatch.setenv("LOCAL_WORLD_SIZE", str(local_world))
    monkeypatch.delenv("PCONV_ENGINE_SPIN_US", raising=False)
    return write


def test_host_cpus_follow_quota_and_local_world(lib, quota_file):
    aff = len(os.sched_getaffinity(0))
    quota_file("max 100000")
    assert lib.pconv_ee_host_cpus() == aff                      # no quota: the affinity mask
    quota_file("1600000 100000")                                # the GPU box: 16 CPUs
    assert lib.pconv_ee_host_cpus() == min(aff, 16)
    quota_file("1600000 100000", local_world=8)                 # ... shared by 8 ranks
    assert lib.pconv_ee_host_cpus() == min(aff, 2)
    quota_file("1600000 100000", local_world=64)                # never below one
    assert lib.pconv_ee_host_cpus() == 1
    quota_file("150000 100000")                                 # 1.5 CPUs -> 1
    assert lib.pconv_ee_host_cpus() == 1


def test_long_poll_only_when_the_calls_threads_fit_the_share(lib, quota_file):
    """the 2 ms poll (through the GPU part of a decoder step) is chosen only when every host thread of the call
    (one per frame) plus the caller has a CPU of this rank's share; otherwise workers block after 60 us"""
    aff = len(os.sched_getaffinity(0))
    quota_file("1600000 100000", local_world=8)                 # 2 CPUs per rank
    share = min(aff, 2)
    for frames in (1, 2, 4, 8):
        assert lib.pconv_ee_spin_us(frames) == (2000 if frames + 1 <= share else 60)
    assert lib.pconv_ee_spin_us(8) == 60                        # the benchmark's 8 frames per rank: never on 2 CPUs
    quota_file("1600000 100000", local_world=1)                 # one rank owns the box's 16 CPUs
    share = min(aff, 16)
    for frames in (1, 2, 4, 8, 16):
        assert lib.pconv_ee_spin_us(frames) == (2000 if frames + 1 <= share else 60)
    quota_file("max 100000")                                    # affinity only
    assert lib.pconv_ee_spin_us(aff) == 60 and lib.pconv_ee_spin_us(max(aff - 1, 0)) == 2000


def test_spin_override(lib, quota_file, monkeypatch):
    quota_file("200000 100000", local_world=2)
    monkeypatch.setenv("PCONV_ENGINE_SPIN_US", "137")
    assert lib.pconv_ee_spin_us(8) == 137 and lib.pconv_ee_spin_us(0) == 137


def test_bench_emulated_rank_is_pinned_to_its_quota_share(quota_file, monkeypatch):
    """bench.py --emulate-local-world N: the one real rank keeps quota / N CPUs of rank 0's affinity slice"""
    sys.path.insert(0, ROOT)
    import bench
    aff = sorted(os.sched_getaffinity(0))
    if len(aff) < 4:
        pytest.skip("needs 4 CPUs")
    pinned = []
    monkeypatch.setattr(os, "sched_setaffinity", lambda pid, cpus: pinned.append(sorted(cpus)))
    monkeypatch.setattr(bench.torch, "set_num_threads", lambda n: None)
    monkeypatch.delenv("PCONV_BENCH_PIN", raising=False)
    quota_file("%d 100000" % (len(aff) * 100000))              # quota == the CPUs we have
    for n in (1, 2, 4):
        cores = bench.pin_rank(0, n, emulate=True)
        assert cores == len(aff) // n
        assert pinned[-1] == aff[:len(aff) // n]
    # a real rank of a real job keeps its whole affinity slice (the quota is shared, not partitioned)
    cores = bench.pin_rank(1, 2)
    assert pinned[-1] == bench.rank_cpus(1, 2) and cores == len(aff) // 2


def _plan(lib, nimg):
    import ctypes
    v = [ctypes.c_int(-1) for _ in range(4)]
    assert lib.pconv_ee_host_plan(nimg, *[ctypes.addressof(x) for x in v]) == 0
    return tuple(x.value for x in v)   # groups, group_threads (0 = one per frame), queued_chain, blocking_sync


def test_host_plan_follows_the_share(lib, quota_file, monkeypatch):
    """one place decides groups / decoding threads / chain / waits (csrc/engine.cpp host_plan).  With CPUs to spare:
    the measured best of round 3.  On a small share (profiles/round5_host_share.txt): never more driver threads
    than CPUs, the driver decodes its group's frames alone, host-driven chain, sleeping waits."""
    for name in ("PCONV_ENGINE_GROUPS", "PCONV_ENGINE_WORKERS", "PCONV_ENGINE_CHAIN", "PCONV_ENGINE_BLOCKING_SYNC"):
        monkeypatch.delenv(name, raising=False)
    aff = len(os.sched_getaffinity(0))
    if aff < 8:
        pytest.skip("needs 8 CPUs in the affinity mask")
    quota_file("1600000 100000", local_world=1)          # 16 CPUs (here: min(aff, 16) >= 8)
    share = min(aff, 16)
    assert _plan(lib, 1) == (1, 0, 1, 0)                 # one frame: one group, queued chain
    assert _plan(lib, 2) == (2, 0, 1, 0)
    assert _plan(lib, 4) == (2, 0, 1, 0)
    if share >= 9:
        assert _plan(lib, 8) == (4, 0, 0, 0)             # the benchmark's 8 frames: four host-driven groups of two
    quota_file("1600000 100000", local_world=4)          # 4 CPUs per rank
    assert _plan(lib, 8) == (4, 1, 0, 1)                 # four drivers, each decodes its two frames itself, waits sleep
    assert _plan(lib, 2) == (2, 0, 1, 0)                 # two frames still fit (2 + 1 <= 4)
    quota_file("1600000 100000", local_world=8)          # 2 CPUs per rank
    assert _plan(lib, 8) == (2, 1, 0, 1)                 # two groups of four
    assert _plan(lib, 2) == (2, 1, 0, 1)                 # 2 + 1 > 2: host-driven, no queueing thread beside the poller
    assert _plan(lib, 1) == (1, 0, 1, 0)
    quota_file("1600000 100000", local_world=16)         # 1 CPU
    assert _plan(lib, 8) == (1, 1, 0, 1)
    # the knobs still win
    monkeypatch.setenv("PCONV_ENGINE_GROUPS", "4")
    monkeypatch.setenv("PCONV_ENGINE_WORKERS", "2")
    monkeypatch.setenv("PCONV_ENGINE_CHAIN", "queued")
    monkeypatch.setenv("PCONV_ENGINE_BLOCKING_SYNC", "0")
    assert _plan(lib, 8) == (4, 2, 1, 0)
