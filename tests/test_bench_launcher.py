"""bench.py's own N > 1 path on the CPU: `python bench.py --gpus 2` must start two
ranks (children, rendezvous on 127.0.0.1), every rank runs `bench.run`, and rank 0
prints one JSON line with n_gpus == 2 and the SUM of both shards.  The ranks use
gloo and a toy oracle workload (tests/helpers/bench_dryrun_rank.py)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "helpers", "bench_dryrun_rank.py")


def _run(nproc, tmp_path, frames=("--frames-per-gpu", "2"), env=None):
    child_env = {"PCONV_DRYRUN_DIR": str(tmp_path)}
    child_env.update(env or {})
    code = ("import sys; sys.path.insert(0, %r); import bench; "
            "sys.exit(bench.launch_ranks(%d, ['--gpus', '%d', '--steps', '2', '--warmup', '1', '--prime', '0', "
            "%r, %r], script=%r, env=%r))"
            % (ROOT, nproc, nproc, frames[0], frames[1], WORKER, child_env))
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


@pytest.mark.timeout(1200)
def test_launcher_starts_two_ranks_and_reduces(tmp_path):
    two = _run(2, tmp_path)
    assert two["n_gpus"] == 2 and two["steps"] == 2 and two["scaling"] == "weak"
    assert two["config"]["frames_per_gpu"] == 2
    # weak scaling: twice the pixels of one rank's shard in the same number of steps
    pixels_per_rank_step = 2 * 256 * 1024
    assert abs(two["value"] * 1e6 * two["ms_per_step"] * 1e-3 - 2 * pixels_per_rank_step) / (2 * pixels_per_rank_step) < 0.01
    assert two["config"]["bpp"] > 0
    # every rank kept its own slice of the host cores (bench.pin_rank), before creating any thread
    allowed = sorted(os.sched_getaffinity(0))
    kept = [json.load(open(os.path.join(str(tmp_path), "affinity_r%d.json" % r))) for r in range(2)]
    if len(allowed) >= 2:
        assert not set(kept[0]["cpus"]) & set(kept[1]["cpus"])
        assert sorted(kept[0]["cpus"] + kept[1]["cpus"]) == allowed
        assert two["config"]["cores_per_rank"] == len(kept[0]["cpus"]) == kept[0]["torch_threads"]


@pytest.mark.timeout(1500)
def test_eight_ranks_on_one_host_share_the_quota(tmp_path):
    """BASELINE config #5's process model (one process per GPU on one host, test/trainDDP_Full.py:83-86,201-204)
    at its full width on the CPU: 8 gloo ranks of bench.run.  Every rank keeps a slice of its own of the host
    CPUs, the native engine of every rank sizes its host threads by quota / 8 (not by the affinity mask), and the
    ranks that would poll through the GPU part of a step never add up to more runnable spinners than the quota."""
    allowed = sorted(os.sched_getaffinity(0))
    quota = 16                                                  # the GPU box's container: 16 CPUs, shared by the ranks
    (tmp_path / "cpu.max").write_text("%d 100000" % (quota * 100000))
    out = _run(8, tmp_path, ("--frames-per-gpu", "1"), env={"PCONV_CGROUP_CPU_MAX": str(tmp_path / "cpu.max")})
    assert out["n_gpus"] == 8 and out["scaling"] == "weak"
    pixels = 8 * 256 * 1024
    assert abs(out["value"] * 1e6 * out["ms_per_step"] * 1e-3 - pixels) / pixels < 0.01
    kept = [json.load(open(os.path.join(str(tmp_path), "affinity_r%d.json" % r))) for r in range(8)]
    assert all(k["local_world"] == 8 for k in kept)
    if len(allowed) >= 8:
        seen = [c for k in kept for c in k["cpus"]]
        assert len(seen) == len(set(seen)) and sorted(seen) == allowed      # disjoint, covering
    for k in kept:
        assert k["engine_host_cpus"] == min(len(k["cpus"]), quota // 8)
        # 8 frames per rank = 8 decode threads + the caller on 2 CPUs: the workers block instead of polling
        assert k["engine_spin_us_8_frames"] == 60
    spinners = sum(8 for k in kept if k["engine_spin_us_8_frames"] > 60)
    assert spinners <= quota


@pytest.mark.timeout(1200)
def test_strong_scaling_mode_splits_a_fixed_batch(tmp_path):
    """--frames-total F (BASELINE config #5: batch 64 over N GPUs): rank r takes frames r::world"""
    two = _run(2, tmp_path, ("--frames-total", "5"))
    assert two["n_gpus"] == 2 and two["scaling"] == "strong" and two["config"]["frames_total"] == 5
    kept = [json.load(open(os.path.join(str(tmp_path), "affinity_r%d.json" % r))) for r in range(2)]
    assert [k["frames"] for k in kept] == [3, 2]
    pixels_step = 5 * 256 * 1024
    assert abs(two["value"] * 1e6 * two["ms_per_step"] * 1e-3 - pixels_step) / pixels_step < 0.01


def test_rank_cpu_slices_are_disjoint_and_cover():
    sys.path.insert(0, ROOT)
    import bench
    for ncpu, world in ((128, 8), (16, 8), (10, 4), (8, 8), (3, 8)):
        allowed = list(range(100, 100 + ncpu))
        parts = [bench.rank_cpus(r, world, allowed) for r in range(world)]
        if ncpu >= world:
            assert sorted(c for p in parts for c in p) == allowed
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
        else:
            assert all(p == allowed for p in parts)


def test_main_becomes_launcher_only_without_world_size(monkeypatch):
    """--gpus N with WORLD_SIZE set (the driver's torch.distributed.run) must NOT spawn again"""
    sys.path.insert(0, ROOT)
    import bench
    calls = []
    monkeypatch.setattr(bench, "launch_ranks", lambda n, argv, **kw: calls.append((n, list(argv))) or 0)
    monkeypatch.setattr(bench, "run", lambda args, **kw: calls.append(("run", args.gpus)))
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "4", "--steps", "1"])
    assert e.value.code == 0 and calls == [(4, ["--gpus", "4", "--steps", "1"])]
    calls.clear()
    monkeypatch.setenv("WORLD_SIZE", "4")
    bench.main(["--gpus", "4", "--steps", "1"])
    assert calls == [("run", 4)]
    calls.clear()
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    bench.main(["--gpus", "1", "--steps", "1"])
    assert calls == [("run", 1)]
