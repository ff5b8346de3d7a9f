"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI; "gloo" in CPU tests).

The path shards by image tile with NO data-path collective: every rank renders its own pixels with absolute
pixel coordinates, so the image is identical for any GPU count.  The only exchange is one gather of the tile-owned
frame-buffer pixels to rank 0 per *output* (not per pass).
"""
from __future__ import annotations

import numpy as np


_INDEX_CACHE = {}


def _device_indices(pixel_lists, r, dev):
    """pixel list r as an int64 index tensor on `dev`, uploaded once per (list, device) — a gather inside a timed region then costs
    the pack, the collective and the scatter only"""
    key = (id(pixel_lists), r, str(dev))
    hit = _INDEX_CACHE.get(key)
    if hit is None or hit[0] is not pixel_lists[r]:
        import torch
        if len(_INDEX_CACHE) > 256:
            _INDEX_CACHE.clear()
        hit = (pixel_lists[r], torch.from_numpy(pixel_lists[r].astype(np.int64)).to(dev))
        _INDEX_CACHE[key] = hit
    return hit[1]


def gather_framebuffer(fb_local, pixel_lists, rank, world_size, dst=0, channels=(5,)):
    """Gather the per-rank owned pixels of the requested channels to `dst`.

    fb_local    : torch tensor (8, n_pixels_full, 4) on this rank's device (only this rank's pixels are meaningful)
    pixel_lists : list (len world_size) of uint32 numpy arrays of absolute pixel indices (tile_pixel_lists)
    returns     : on dst, a tensor (len(channels), n_pixels_full, 4) holding every rank's pixels; None elsewhere.
    Message size: 16 B x len(channels) x n/world_size per rank — e.g. 2.9 MB per rank for COMPOSITED_C at 1600x900 on 8 GPUs.
    """
    import torch
    import torch.distributed as dist
    dev = fb_local.device
    ch = list(channels)
    # gloo (CPU tests, single-GPU dry runs of the N>1 path) has no device collectives: stage through host memory there
    on_host = world_size > 1 and dist.get_backend() == "gloo" and dev.type != "cpu"
    mine = _device_indices(pixel_lists, rank, dev)
    packed = fb_local[ch][:, mine, :].contiguous()                       # (C, n_local, 4)
    if world_size == 1:
        out = torch.zeros((len(ch),) + tuple(fb_local.shape[1:]), dtype=fb_local.dtype, device=dev)
        out[:, mine, :] = packed
        return out
    # ranks own different pixel counts: pad to the maximum so one fixed-size gather suffices
    n_max = max(len(p) for p in pixel_lists)
    buf = torch.zeros((len(ch), n_max, 4), dtype=fb_local.dtype, device=dev)
    buf[:, :packed.shape[1], :] = packed
    if on_host:
        buf = buf.cpu()
    if rank == dst:
        recv = [torch.zeros_like(buf) for _ in range(world_size)]
        dist.gather(buf, recv, dst=dst)
        out = torch.zeros((len(ch),) + tuple(fb_local.shape[1:]), dtype=fb_local.dtype, device=dev)
        for r in range(world_size):
            out[:, _device_indices(pixel_lists, r, dev), :] = recv[r][:, :len(pixel_lists[r]), :].to(dev)
        return out
    dist.gather(buf, None, dst=dst)
    return None


FILTER_INPUT_CHANNELS = (0, 1, 2, 3, 4)      # DIFFUSE_C, DIFFUSE_A, SPECULAR_C, SPECULAR_A, DIRECT_C


def gather_filter_inputs(fb_local, gb_geo_local, pixel_lists, rank, world_size, dst=0):
    """Collect what RenderingContextImpl::filter reads (src/renderer.cu:1099-1151) on `dst`: the five input channels and the
    gbuffer geometry of every rank's tiles.  The 7-step a-trous filter reaches 2*(1+2+...+64) = 254 pixels, i.e. across every
    tile boundary, so it is run on the assembled frame (one gather, 6 x 16 B x n/world_size per rank: 17 MB per rank at
    1600x900 on 8 GPUs; the filter itself is ~0.3 ms on one MI355X).

    returns on dst: (fb_full, gb_geo_full) with fb_full (8, n, 4) holding the gathered channels (others zero); (None, None) elsewhere.
    """
    import torch
    n = fb_local.shape[1]
    ext = torch.cat([fb_local[list(FILTER_INPUT_CHANNELS)], gb_geo_local.reshape(1, n, 4).to(fb_local.dtype)], 0)      # bit patterns travel untouched
    out = gather_framebuffer(ext, pixel_lists, rank, world_size, dst=dst, channels=tuple(range(len(FILTER_INPUT_CHANNELS) + 1)))
    if out is None:
        return None, None
    fb_full = torch.zeros((8, n, 4), dtype=fb_local.dtype, device=out.device)
    fb_full[list(FILTER_INPUT_CHANNELS)] = out[:len(FILTER_INPUT_CHANNELS)]
    return fb_full, out[len(FILTER_INPUT_CHANNELS)].contiguous()


def allreduce_splats(splats, world_size):
    """Bidirectional path tracer under tile sharding: every rank's light sub-paths splat onto ARBITRARY pixels, so the
    per-pixel light-tracing sums (int64 2^-32 fixed point, 3 per pixel; order-independent by construction) are summed over the
    ranks with one integer all-reduce (RCCL over xGMI: 24 B x n pixels = 34.6 MB at 1600x900 per pass; with passes in flight the
    buffer holds one such slab per pass of the batch and is reduced once per batch) before each rank folds them into its frame.
    In place; a no-op on one rank."""
    if world_size == 1:
        return splats
    import torch.distributed as dist
    if dist.get_backend() == "gloo" and splats.device.type != "cpu":
        host = splats.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        splats.copy_(host)
    else:
        dist.all_reduce(splats, op=dist.ReduceOp.SUM)
    return splats
