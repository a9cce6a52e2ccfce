"""The N > 1 path on CPU: two ranks (gloo), independent chains per rank, one all-gather of the
retained allocation samples, posterior-similarity rows split across ranks."""
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
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as G
    G.load_package()
    from particlemdi_jl_amd.psm import allgather_samples, psm_rows
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    T, C, K, n = 3, 2, 2, 10
    rng = np.random.default_rng(100 + rank)                       # chains differ per rank (seeds 100..)
    mine = torch.from_numpy(rng.integers(1, 4, size=(T, C, K, n)).astype(np.uint8))
    pooled = allgather_samples(mine)                              # the one collective of the path
    assert pooled.shape == (world * T * C, K, n)
    lo, hi = rank * n // world, (rank + 1) * n // world           # PSM rows are partitioned, no further exchange
    rows = psm_rows(pooled, lo, hi, host=True)
    np.save(os.path.join(outdir, f"rows{rank}.npy"), rows.numpy())
    np.save(os.path.join(outdir, f"mine{rank}.npy"), mine.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_allgather_and_psm_two_ranks(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    import __graft_entry__ as G
    G.load_package()
    from particlemdi_jl_amd.psm import psm_rows
    mine = [np.load(tmp_path / f"mine{r}.npy") for r in range(world)]
    pooled = np.concatenate([m.reshape(-1, m.shape[-2], m.shape[-1]) for m in mine])
    want = psm_rows(pooled, 0, pooled.shape[-1], host=True)
    got = np.concatenate([np.load(tmp_path / f"rows{r}.npy") for r in range(world)], axis=1)
    assert np.allclose(got, want)
