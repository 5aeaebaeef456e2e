"""N > 1: the frame partition and the framebuffer merge, with world_size 2 over gloo on the CPU.
The per-rank renderer here is the CPU oracle (this container has no GPU); on the GPU the same
partition (prt_set_row_blocks) and the same merge (parallel.merge_on_rank0 -> one all-gather) carry
libprt's framebuffer, which tests/test_gpu_parity.py checks tile by tile."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG_NAME, ROOT

W, H, FRAMES, WORLD = 48, 41, 40, 2


def _worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    prt = importlib.import_module(PKG_NAME)
    par = importlib.import_module(PKG_NAME + ".parallel")
    import oracle_api as O
    scene = prt.HostScene("cornell_coat.json")
    cfg = scene.config()
    cam = prt.default_camera(W, H)
    seeds = prt.seed_pairs(FRAMES)
    rows = par.rows_of_rank(H, world, rank)
    _, tile = O.Restatement().render(cfg, scene.desc, cam, W, H, seeds, blocks=(par.BLOCK_ROWS, world, rank), threads=2)
    assert tile.shape[0] == len(rows)
    tile[0, 0, 3] = -0.0                                   # a bit pattern a sum of zero-padded frames would lose
    padded = torch.zeros((par.max_rows_per_rank(H, world), W, 4), dtype=torch.float32)
    padded[:len(rows)] = torch.from_numpy(tile)
    full = par.merge_on_rank0(padded, H, W, world, dist)
    dist.barrier()
    if rank == 0:
        np.save(out_path, full.numpy())
    dist.destroy_process_group()


def test_two_ranks_reproduce_the_single_rank_image(prt, oracle, tmp_path):
    par = importlib.import_module(PKG_NAME + ".parallel")
    r0, r1 = par.rows_of_rank(H, 2, 0), par.rows_of_rank(H, 2, 1)
    assert sorted(np.concatenate([r0, r1]).tolist()) == list(range(H)) and abs(len(r0) - len(r1)) <= par.BLOCK_ROWS
    out = str(tmp_path / "merged.npy")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(WORLD, port, out), nprocs=WORLD, join=True)
    merged = np.load(out)
    scene = prt.HostScene("cornell_coat.json")
    _, single = oracle.Restatement().render(scene.config(), scene.desc, prt.default_camera(W, H), W, H, prt.seed_pairs(FRAMES), threads=4)
    par = importlib.import_module(PKG_NAME + ".parallel")
    for rk in range(WORLD):
        single[par.rows_of_rank(H, WORLD, rk)[0], 0, 3] = -0.0        # what the workers planted
    assert np.array_equal(single.view(np.uint32), merged.view(np.uint32))   # the rows themselves travel: every bit pattern survives
