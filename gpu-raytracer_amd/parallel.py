"""Multi-GPU frame split: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).

The reference is single-GPU; what it does have is batch rendering of contiguous scan-order pixel
ranges (kernel_generate(sample, pixel_offset, pixel_count), Pathtracer.cu:122-131). Paths of
different pixels are independent and the RNG is keyed on the pixel index (Sampling.h:46,71-72),
so any partition of the frame gives bit-identical pixels. Each rank therefore renders a disjoint
set of row tiles with a full scene replica, and the only data-path collective per frame is one
all-gather of the final float4 image (33 MB at 1080p, 4 MB per rank at 8 GPUs).

Tiles are dealt round-robin (rank r gets tiles r, r+W, r+2W, ...) instead of one contiguous block
per rank: Sponza's upper rows see sky and terminate early, so contiguous blocks would leave the
ranks holding the floor rows as stragglers.
"""
import numpy as np

TILE_ROWS = 8


def tile_layout(width, height, world_size, tile_rows=TILE_ROWS):
    """Returns (tile_pixels, tiles_total, padded_tiles): the frame is cut into tiles of
    `tile_rows` full rows; the tile count is padded up to a multiple of world_size so every
    rank owns the same number of tiles (padding tiles lie below the frame and are skipped)."""
    tiles_total = (height + tile_rows - 1) // tile_rows
    padded = (tiles_total + world_size - 1) // world_size * world_size
    return width * tile_rows, tiles_total, padded


def rank_tiles(rank, world_size, width, height, tile_rows=TILE_ROWS):
    """Scan-order pixel ranges [(offset, count), ...] owned by `rank`."""
    tile_pixels, tiles_total, _ = tile_layout(width, height, world_size, tile_rows)
    frame_pixels = width * height
    ranges = []
    for tile in range(rank, tiles_total, world_size):
        offset = tile * tile_pixels
        ranges.append((offset, min(tile_pixels, frame_pixels - offset)))
    return ranges


def gather_order(world_size, width, height, tile_rows=TILE_ROWS):
    """Index array mapping the all-gathered buffer [world, tiles_per_rank, tile_pixels] back to
    scan order: full[p] = gathered.reshape(-1)[order[p]] for every frame pixel p."""
    tile_pixels, tiles_total, padded = tile_layout(width, height, world_size, tile_rows)
    per_rank = padded // world_size
    frame_pixels = width * height
    p = np.arange(frame_pixels)
    tile = p // tile_pixels
    owner, slot = tile % world_size, tile // world_size
    return (owner * per_rank + slot) * tile_pixels + p % tile_pixels


class TileSplit:
    """Per-rank state for rendering + gathering one frame with torch.distributed."""

    def __init__(self, rank, world_size, width, height, tile_rows=TILE_ROWS):
        self.rank, self.world_size, self.width, self.height, self.tile_rows = rank, world_size, width, height, tile_rows
        self.tile_pixels, self.tiles_total, self.padded_tiles = tile_layout(width, height, world_size, tile_rows)
        self.tiles_per_rank = self.padded_tiles // world_size
        self.ranges = rank_tiles(rank, world_size, width, height, tile_rows)
        self.local_pixels = self.tiles_per_rank * self.tile_pixels

    def pack(self, image_rows):
        """image_rows: array/tensor [height, pitch>=width, C]; returns this rank's tiles packed
        as [tiles_per_rank * tile_pixels, C] (zero padded)."""
        import torch
        t = torch.as_tensor(image_rows)
        flat = t[:, :self.width, :].reshape(-1, t.shape[-1])
        out = torch.zeros((self.local_pixels, t.shape[-1]), dtype=t.dtype, device=t.device)
        for slot, (offset, count) in enumerate(self.ranges):
            out[slot * self.tile_pixels: slot * self.tile_pixels + count] = flat[offset:offset + count]
        return out

    def unpack(self, gathered):
        """gathered: [world * local_pixels, C] from all_gather_into_tensor -> [height, width, C]."""
        import torch
        order = torch.as_tensor(gather_order(self.world_size, self.width, self.height, self.tile_rows), device=gathered.device)
        return gathered.reshape(-1, gathered.shape[-1])[order].reshape(self.height, self.width, gathered.shape[-1])

    def all_gather(self, packed):
        import torch
        import torch.distributed as dist
        out = torch.empty((self.world_size * self.local_pixels, packed.shape[-1]), dtype=packed.dtype, device=packed.device)
        if self.world_size == 1:
            out.copy_(packed)
        else:
            dist.all_gather_into_tensor(out, packed.contiguous())
        return out
