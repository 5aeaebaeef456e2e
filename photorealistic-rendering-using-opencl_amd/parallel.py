"""Multi-GPU plumbing: how the frame is partitioned over ranks and how the pieces are merged.

The path shards with no data-path exchange: every pixel is independent (its state, RNG keying and
output depend only on its own global coordinates and the frame number), so each rank renders its
own rows and ONE collective at the end puts the framebuffer together on rank 0 -- a sum of
zero-padded full-size buffers (`torch.distributed.reduce`, RCCL over xGMI when the backend is nccl).
Rows are dealt in interleaved blocks of 16 (prt_set_row_blocks) so the ranks get equal shares of
the expensive middle of the picture.
"""
import numpy as np

BLOCK_ROWS = 16


def rows_of_rank(height, world, rank, block=BLOCK_ROWS):
    """global row indices owned by `rank`, in local order (matches prt_set_row_blocks)"""
    rows = np.arange(height)
    return rows[(rows // block) % world == rank]


def merge_on_rank0(tile, rows, height, width, dist, device=None):
    """tile: (len(rows), width, 4) float32 torch tensor with this rank's rows.  Returns the full
    (height, width, 4) tensor; only rank 0's copy is complete (reduce, not all-reduce)."""
    import torch
    full = torch.zeros((height, width, 4), dtype=torch.float32, device=tile.device if device is None else device)
    idx = torch.as_tensor(np.asarray(rows), dtype=torch.long, device=full.device)
    full.index_copy_(0, idx, tile.to(full.device))
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(full, dst=0, op=dist.ReduceOp.SUM)
    return full
