"""Frame partition for multi-GPU runs, host side.

The frame is cut into square tiles dealt round-robin: tile t (row-major over the tile grid) belongs to rank
t % n_ranks and is that rank's slot t // n_ranks.  A rank's buffer holds its slots back to back, each as
tile x tile RGB pixels; pixels of edge tiles that fall outside the frame, and the unused slots of a short last
round, are zero.  `rt_render_tiles` fills such a buffer on the GPU, `rt_assemble_tiles` scatters the gathered
buffers of all ranks back into a row-major frame.  The numpy functions here restate that contract for tests and
for hosts without a GPU; `TileGather` is the one exchange step of the pipeline (a gather of disjoint tiles to
rank 0, no reduction), used by bench.py over RCCL and by the CPU tests over gloo.
"""
import numpy as np


def grid(width, height, tile):
    return (width + tile - 1) // tile, (height + tile - 1) // tile


def tile_count(width, height, tile):
    tx, ty = grid(width, height, tile)
    return tx * ty


def tiles_per_rank(width, height, tile, n_ranks):
    return (tile_count(width, height, tile) + n_ranks - 1) // n_ranks


def rank_tiles(width, height, tile, n_ranks, rank):
    """Tile ids owned by `rank`, in slot order."""
    return list(range(rank, tile_count(width, height, tile), n_ranks))


def pack_rank_tiles(frame, tile, n_ranks, rank):
    """What a rank's tile buffer holds for a given full frame (H, W, 3) uint8."""
    h, w, _ = frame.shape
    tx, _ = grid(w, h, tile)
    out = np.zeros((tiles_per_rank(w, h, tile, n_ranks), tile, tile, 3), dtype=np.uint8)
    for slot, t in enumerate(rank_tiles(w, h, tile, n_ranks, rank)):
        y0, x0 = (t // tx) * tile, (t % tx) * tile
        part = frame[y0:y0 + tile, x0:x0 + tile]
        out[slot, :part.shape[0], :part.shape[1]] = part
    return out.reshape(-1)


def assemble(gathered, width, height, tile, n_ranks):
    """Row-major frame from the rank-major concatenation of all ranks' tile buffers."""
    tx, _ = grid(width, height, tile)
    per_rank = tiles_per_rank(width, height, tile, n_ranks)
    bufs = np.asarray(gathered, dtype=np.uint8).reshape(n_ranks, per_rank, tile, tile, 3)
    frame = np.zeros((height, width, 3), dtype=np.uint8)
    for t in range(tile_count(width, height, tile)):
        y0, x0 = (t // tx) * tile, (t % tx) * tile
        hh, ww = min(tile, height - y0), min(tile, width - x0)
        frame[y0:y0 + hh, x0:x0 + ww] = bufs[t % n_ranks, t // n_ranks, :hh, :ww]
    return frame


class TileGather:
    """Buffers and the collective for one frame's tiles: every rank contributes `local`, rank 0 receives all of them
    rank-major in `gathered`.  Works on any torch device; `dist` is torch.distributed (initialised) or None."""

    def __init__(self, torch, dist, width, height, tile, n_ranks, rank, device):
        self.dist, self.rank, self.n_ranks = dist, rank, n_ranks
        self.per_rank = tiles_per_rank(width, height, tile, n_ranks) * tile * tile * 3
        self.local = torch.zeros(self.per_rank, dtype=torch.uint8, device=device)
        self.gathered = torch.zeros(n_ranks * self.per_rank, dtype=torch.uint8, device=device) if rank == 0 else None
        self._views = list(self.gathered.split(self.per_rank)) if rank == 0 else None

    def gather(self):
        if self.n_ranks == 1 or self.dist is None:
            if self.rank == 0:
                self.gathered.copy_(self.local)
            return self.gathered
        if self.local.is_cuda and self.dist.get_backend() == "gloo":
            # rehearsal transport (several ranks sharing one GPU): stage through host memory
            host = self.local.cpu()
            parts = [host.new_empty(host.shape) for _ in range(self.n_ranks)] if self.rank == 0 else None
            self.dist.gather(host, parts, dst=0)
            if self.rank == 0:
                for view, part in zip(self._views, parts):
                    view.copy_(part)
            return self.gathered
        self.dist.gather(self.local, self._views, dst=0)
        return self.gathered
