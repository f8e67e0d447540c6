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


class SvgfTileSplit(TileSplit):
    """SVGF / TAA frames under the tile split (BASELINE config 3 on N GPUs). The filter stage needs
    neighbourhoods of +-(3 + 2^5) pixels and reprojects from anywhere in the previous frame, so every rank filters the whole
    frame -- redundantly, but without a second exchange: per frame a rank path-traces its own tiles, ONE all-gather moves what
    the filter reads of this frame (DIRECT / INDIRECT / ALBEDO and the three g-buffers: 5 float4 = 80 B per pixel, 166 MB at
    1080p), and every rank runs reproject / variance / a-trous / finalize / TAA on the full frame."""

    FLOATS_PER_PIXEL = 20

    def __init__(self, rank, world_size, width, height, tile_rows=TILE_ROWS):
        super().__init__(rank, world_size, width, height, tile_rows)
        self._buffers = None

    def render_frame(self, grt, ctx, sample_index, gather=None):
        """One filtered frame on the device context `ctx` (tiles already set with rt_set_pixel_tiles). `gather(packed) ->
        gathered` defaults to torch.distributed.all_gather_into_tensor; it is ordered on torch's current stream."""
        import ctypes
        import torch
        lib = grt.device_lib()
        for name, args in (("rt_render_sample_unfiltered", [ctypes.c_void_p, ctypes.c_int]), ("rt_filter_frame", [ctypes.c_void_p, ctypes.c_int]),
                           ("rt_pack_svgf_inputs", [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 4), ("rt_unpack_svgf_inputs", [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 3),
                           ("rt_stream_wait_for_context", [ctypes.c_void_p, ctypes.c_void_p]), ("rt_context_wait_for_stream", [ctypes.c_void_p, ctypes.c_void_p])):
            getattr(lib, name).argtypes = args

        def check(status):
            if status != 0:
                raise RuntimeError(lib.rt_last_error(ctx).decode())

        if self._buffers is None:
            device = torch.device("cuda", torch.cuda.current_device())
            self._buffers = (torch.zeros((self.local_pixels, self.FLOATS_PER_PIXEL), dtype=torch.float32, device=device),
                             torch.zeros((self.world_size * self.local_pixels, self.FLOATS_PER_PIXEL), dtype=torch.float32, device=device))
        packed, gathered = self._buffers
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(lib.rt_render_sample_unfiltered(ctx, sample_index))
        check(lib.rt_context_wait_for_stream(ctx, stream))      # the previous frame's collective has read `packed`
        check(lib.rt_pack_svgf_inputs(ctx, packed.data_ptr(), self.tile_pixels, self.rank, self.world_size, self.tiles_per_rank))
        check(lib.rt_stream_wait_for_context(ctx, stream))
        if gather is not None:
            gathered = gather(packed)
        elif self.world_size == 1:
            gathered.copy_(packed)
        else:
            import torch.distributed as dist
            dist.all_gather_into_tensor(gathered, packed)
        check(lib.rt_context_wait_for_stream(ctx, stream))
        check(lib.rt_unpack_svgf_inputs(ctx, gathered.data_ptr(), self.tile_pixels, self.world_size, self.tiles_per_rank))
        check(lib.rt_filter_frame(ctx, sample_index))
