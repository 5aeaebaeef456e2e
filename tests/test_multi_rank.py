"""N > 1: the frame partition and the framebuffer merge, with world_size 2 and 4 over gloo on the CPU.
The per-rank renderer here is the CPU oracle (this container has no GPU); on the GPU the same
partition (prt_set_row_blocks) and the same merge (parallel.merge_on_rank0 -> one gather to rank 0) carry
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

W, FRAMES = 48, 40
# world 2: 41 rows = 16 + 16 + 9 (the last block is short and goes to rank 0); world 4: 73 rows = 4 x 16 + 9: rank 0 owns 25 rows, the
# others 16 -- the tiles are padded to the longest share and the padding must not reach the merged frame
CASES = {2: 41, 4: 73}


def _worker(rank, world, port, out_path, H):
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
    assert (full is None) == (rank != 0)                   # a gather: only rank 0 holds the frame
    if rank == 0:
        np.save(out_path, full.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_ranks_reproduce_the_single_rank_image(prt, oracle, tmp_path, world):
    par = importlib.import_module(PKG_NAME + ".parallel")
    H = CASES[world]
    shares = [par.rows_of_rank(H, world, r) for r in range(world)]
    assert sorted(np.concatenate(shares).tolist()) == list(range(H)) and all(len(s) for s in shares)
    assert len({len(s) for s in shares}) > 1                              # uneven shares: some tiles carry padding rows
    out = str(tmp_path / "merged.npy")
    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, out, H), nprocs=world, join=True)
    merged = np.load(out)
    scene = prt.HostScene("cornell_coat.json")
    _, single = oracle.Restatement().render(scene.config(), scene.desc, prt.default_camera(W, H), W, H, prt.seed_pairs(FRAMES), threads=4)
    for rk in range(world):
        single[shares[rk][0], 0, 3] = -0.0                                # what the workers planted
    assert np.array_equal(single.view(np.uint32), merged.view(np.uint32))   # the rows themselves travel: every bit pattern survives
