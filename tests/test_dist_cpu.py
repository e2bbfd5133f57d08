"""N > 1 path on CPU: two gloo ranks each code their shard of frames with the
oracle backend (no data-path collective) and reduce their metric sums; the totals
must equal a single-process run over all frames."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _code_frames(indices, tmpdir):
    """encode+decode the given frame indices with the oracle; returns metric sums"""
    from pseudocylindrical_convolution_amd.PCONV_operator import backend
    from oracle import pconv_cpu, coder_cpu
    backend.use(pconv_cpu, coder_cpu)
    pconv_cpu.set_detmath(True)
    from pseudocylindrical_convolution_amd import pseudo_codec as PC
    torch.manual_seed(1234)
    ent_e = PC.EntEncoder(4, 16, True, 8, gid=0)
    ent_d = PC.EntDecoder(4, 16, True, 8, gid=0)
    g = torch.Generator().manual_seed(7)
    sd = {k: torch.randn(v.shape, generator=g) * 0.05 for k, v in ent_e.state_dict().items()}
    ent_e.load_state_dict(sd)
    ent_d.load_state_dict(sd)
    out = {"pixels": 0.0, "bits": 0.0, "frames": 0.0}
    for i in indices:
        sym = torch.randint(0, 8, (16, 4, 1, 64), generator=torch.Generator().manual_seed(100 + i)).float()
        path = os.path.join(tmpdir, "f%d.bin" % i)
        ent_e.start(path)
        ent_e(sym.clone())
        ent_d.start(path)
        dec = ent_d(1, 64)
        assert torch.equal(dec, ent_e.fill(sym.clone()))
        out["pixels"] += 128 * 512.0
        out["bits"] += os.path.getsize(path) * 8.0
        out["frames"] += 1
    return out


def _worker(rank, world, port, tmpdir, total, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pseudocylindrical_convolution_amd import sharding
    mine = sharding.shard(total, rank, world)
    local = _code_frames(mine, tmpdir)
    totals, secs = sharding.reduce_metrics(local, seconds=1.0 + rank)
    ret[rank] = (mine, totals, secs)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_shard_frames_and_reduce_metrics(tmp_path):
    from pseudocylindrical_convolution_amd import sharding
    total, world = 4, 2
    assert sharding.shard(5, 1, 2) == [1, 3] and sharding.shard(5, 0, 2) == [0, 2, 4]
    single = _code_frames(range(total), str(tmp_path))
    from pseudocylindrical_convolution_amd.PCONV_operator import backend
    backend.reset()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), total, ret), nprocs=world, join=True)
    assert sorted(ret[0][0] + ret[1][0]) == list(range(total))
    for r in range(world):
        totals, secs = ret[r][1], ret[r][2]
        assert secs == 2.0                                         # MAX over ranks
        for k in ("pixels", "bits", "frames"):
            assert totals[k] == single[k], k                        # SUM over ranks == one process


def test_reduce_metrics_without_process_group():
    from pseudocylindrical_convolution_amd import sharding
    totals, secs = sharding.reduce_metrics({"pixels": 10, "bits": 3}, 0.5)
    assert totals["pixels"] == 10 and totals["bits"] == 3 and totals["frames"] == 0 and secs == 0.5
