"""Multi-process (world_size 2, gloo, CPU) test of the N > 1 path: batch sharding + optional coefficient
gather.  The per-rank transform runs on the oracle-backed test engine here (no GPU in this tier); on the GPU
box the same code path runs with the HIP engine under torchrun (bench.py --gpus N)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, fn_name, batch, tmp):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ptwt_amd
        from ptwt_amd import _engine, distributed as D
        from tests._oracle_engine import OracleLevelEngine

        _engine.ENGINE = OracleLevelEngine()  # CPU stand-in for the HIP level engine (tests only)
        g = torch.Generator().manual_seed(42)
        shape = (batch, 20, 22) if fn_name != "wavedec" else (batch, 50)
        full = torch.randn(*shape, generator=g, dtype=torch.float64)
        local = D.shard_batch(full)  # this rank's contiguous slice of the batch
        lo, hi = D.shard_bounds(batch, rank, world)
        assert local.shape[0] == hi - lo
        fn = getattr(ptwt_amd, fn_name)
        coeffs_local = fn(local, "db2", level=2)  # no collective in the transform
        gathered = D.gather_coeffs(coeffs_local)
        want = fn(full, "db2", level=2)
        assert type(gathered) is type(want)
        flat_g = [t for c in gathered for t in (c.values() if isinstance(c, dict) else (c if isinstance(c, tuple) else [c]))]
        flat_w = [t for c in want for t in (c.values() if isinstance(c, dict) else (c if isinstance(c, tuple) else [c]))]
        assert len(flat_g) == len(flat_w)
        for a, b in zip(flat_g, flat_w):
            assert a.shape == b.shape and torch.equal(a, b)
        dist.barrier()
        open(os.path.join(tmp, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("fn_name,batch", [("wavedec2", 6), ("wavedec2", 5), ("fswavedec2", 3), ("wavedec", 4)])
def test_sharded_transform_and_gather_world2(tmp_path, fn_name, batch):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), fn_name, batch, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def test_shard_bounds():
    from ptwt_amd.distributed import shard_bounds

    assert [shard_bounds(512, r, 8) for r in range(8)] == [(64 * r, 64 * r + 64) for r in range(8)]
    assert [shard_bounds(5, r, 2) for r in range(2)] == [(0, 3), (3, 5)]
    assert [shard_bounds(2, r, 4) for r in range(4)] == [(0, 1), (1, 2), (2, 2), (2, 2)]
    with pytest.raises(ValueError):
        shard_bounds(4, 4, 4)
