"""Multi-GPU plumbing: how the frame is partitioned over ranks and how the pieces are merged.

The path shards with no data-path exchange: every pixel is independent (its state, RNG keying and
output depend only on its own global coordinates and the frame number), so each rank renders its
own rows and ONE collective at the end puts the framebuffer together on rank 0: a gather of
the ranks' row tiles to rank 0 (RCCL over xGMI when the backend is nccl), scattered into place by row index.
Moving the rows themselves is lossless for every bit pattern (a sum of zero-padded frames would turn
-0.0 into +0.0) and moves each pixel once instead of N times.
Rows are dealt in interleaved blocks of 16 (prt_set_row_blocks) so the ranks get equal shares of
the expensive middle of the picture.
"""
import numpy as np

BLOCK_ROWS = 16


def rows_of_rank(height, world, rank, block=BLOCK_ROWS):
    """global row indices owned by `rank`, in local order (matches prt_set_row_blocks)"""
    rows = np.arange(height)
    return rows[(rows // block) % world == rank]


def max_rows_per_rank(height, world, block=BLOCK_ROWS):
    return max(len(rows_of_rank(height, world, r, block)) for r in range(world))


def merge_on_rank0(tile, height, width, world, dist):
    """tile: (max_rows_per_rank, width, 4) float32 torch tensor, this rank's rows first (padding rows are ignored).
    ONE collective: a gather to rank 0 (every pixel crosses a link once).  Returns the full (height, width, 4) tensor on rank 0,
    None on the other ranks."""
    import torch
    if world == 1 or dist is None or not dist.is_initialized():
        return tile[:height]
    rank = dist.get_rank()
    pieces = [torch.empty_like(tile) for _ in range(world)] if rank == 0 else None
    dist.gather(tile.contiguous(), gather_list=pieces, dst=0)
    if rank != 0:
        return None
    full = torch.empty((height, width, 4), dtype=torch.float32, device=tile.device)
    for r in range(world):
        rows = rows_of_rank(height, world, r)
        idx = torch.as_tensor(np.asarray(rows), dtype=torch.long, device=tile.device)
        full.index_copy_(0, idx, pieces[r][:len(rows)])
    return full
