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


# ---- the C-ABI route (include/fermat_pt_hip.h "multi-GPU"): the communicator and the gather live inside libfermat_pt_hip.so, on the
#      library's own stream; torch.distributed only carries the 128-byte RCCL id to the other ranks --------------------------------
def comm_init(renderer, rank, world_size):
    """create this rank's RCCL communicator inside the library (fpt_comm_unique_id on rank 0 -> broadcast -> fpt_comm_init)"""
    import ctypes as C
    L = renderer.L
    buf = C.create_string_buffer(128)
    if rank == 0 and L.fpt_comm_unique_id(buf) != 0:
        L.fpt_comm_last_error.restype = C.c_char_p
        raise RuntimeError("fpt_comm_unique_id: " + L.fpt_comm_last_error().decode())
    obj = [bytes(buf.raw) if rank == 0 else None]
    if world_size > 1:
        import torch.distributed as dist
        dist.broadcast_object_list(obj, src=0)
    renderer._check(L.fpt_comm_init(renderer.ctx, C.c_int(rank), C.c_int(world_size), C.c_char_p(obj[0])))


def comm_info(renderer):
    """(rank, world size) of the library's communicator as RCCL itself reports them (ncclCommUserRank / ncclCommCount)"""
    import ctypes as C
    r, w = C.c_int(-1), C.c_int(0)
    renderer._check(renderer.L.fpt_comm_info(renderer.ctx, C.byref(r), C.byref(w)))
    return r.value, w.value


def _channel_mask(channels):
    mask = 0
    for c in channels:
        mask |= 1 << c
    return mask


def set_tile_lists(renderer, pixel_lists, rank, root=0):
    """fpt_set_tile_lists: register the ranks' pixel lists once (device copies: this rank's list, on the root everybody's)"""
    import ctypes as C
    W = len(pixel_lists)
    lists = [np.ascontiguousarray(p, np.uint32) for p in pixel_lists]
    ptrs = (C.c_void_p * W)(*[p.ctypes.data for p in lists])
    counts = (C.c_uint32 * W)(*[len(p) for p in lists])
    renderer._check(renderer.L.fpt_set_tile_lists(renderer.ctx, C.c_int(rank), C.c_int(W), C.c_int(root), ptrs, counts))


def gather_pack(renderer, channels=(5,)):
    """fpt_gather_pack: this rank's owned pixels of `channels` as one message; returns (device pointer, float count) -- valid until the next pack / gather"""
    import ctypes as C
    ptr = C.c_void_p(0); n = C.c_uint64(0)
    renderer._check(renderer.L.fpt_gather_pack(renderer.ctx, C.byref(renderer.view), C.c_uint32(_channel_mask(channels)), C.byref(ptr), C.byref(n)))
    return ptr.value, int(n.value)


def gather_unpack(renderer, src_rank, d_message, channels=(5,)):
    """fpt_gather_unpack: scatter the message rank `src_rank` packed (device pointer on this renderer's device) into renderer.fb"""
    import ctypes as C
    renderer._check(renderer.L.fpt_gather_unpack(renderer.ctx, C.byref(renderer.view), C.c_uint32(_channel_mask(channels)), C.c_int(src_rank), C.c_void_p(d_message)))


def gather_framebuffer_capi(renderer, pixel_lists, root=0, channels=(5,)):
    """fpt_gather_framebuffer: completes the requested channels of `renderer.fb` in place on `root` (grouped ncclSend / ncclRecv on the
    library's stream; asynchronous -- call renderer.synchronize() before reading)"""
    import ctypes as C
    mask = _channel_mask(channels)
    if pixel_lists is not None:
        W = len(pixel_lists)
        lists = [np.ascontiguousarray(p, np.uint32) for p in pixel_lists]
        ptrs = (C.c_void_p * W)(*[p.ctypes.data for p in lists])
        counts = (C.c_uint32 * W)(*[len(p) for p in lists])
    if pixel_lists is None:          # the tables set_tile_lists registered
        renderer._check(renderer.L.fpt_gather_framebuffer(renderer.ctx, C.byref(renderer.view), C.c_int(root), C.c_uint32(mask), None, None))
        return
    renderer._check(renderer.L.fpt_gather_framebuffer(renderer.ctx, C.byref(renderer.view), C.c_int(root), C.c_uint32(mask), ptrs, counts))
    renderer._keep_lists = lists


def allreduce_splats_capi(renderer):
    """fpt_bpt_allreduce_splats over the renderer's deferred splat buffer (bpt_defer_splats)"""
    import ctypes as C
    renderer._check(renderer.L.fpt_bpt_allreduce_splats(renderer.ctx, C.c_uint64(renderer.splats.numel())))


PSF_RECORD_BYTES = 40


def device_bytes(ptr, nbytes, dev):
    """a uint8 torch tensor over `nbytes` of device memory the library owns (no copy; valid as long as the library keeps the buffer)"""
    import torch

    class _Raw:
        __cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(_Raw(), device=dev)


def exchange_psf_cells(renderer, rank, world_size):
    """PSFPT under tile sharding, the torch.distributed route (gloo tests, single-GPU dry runs): all-gather every rank's records of the pending
    pass (Renderer.psf_export_cells) and merge all of them, the own ones included, into this rank's global table.  On distinct GPUs with the
    nccl backend use Renderer.psf_exchange_cells instead (RCCL inside the library, no host staging)."""
    import torch
    import torch.distributed as dist
    ptr, n = renderer.psf_export_cells()
    mine = torch.empty(0, dtype=torch.uint8)
    if n:
        mine = device_bytes(ptr, n * PSF_RECORD_BYTES, renderer.dev).cpu()
    lists = [None] * world_size
    if world_size > 1:
        dist.all_gather_object(lists, mine.numpy().tobytes())
    else:
        lists = [mine.numpy().tobytes()]
    keep = []
    for r in range(world_size):
        b = lists[r]
        if not b:
            continue
        t = torch.frombuffer(bytearray(b), dtype=torch.uint8).to(renderer.dev)
        keep.append(t)
        renderer.psf_import_cells(t.data_ptr(), len(b) // PSF_RECORD_BYTES)
    renderer.synchronize()          # the staged tensors must outlive the merge kernels
    return sum(len(b) for b in lists if b) // PSF_RECORD_BYTES


# ---- BPT -sc 1 with shared light vertices (include/fermat_pt_hip.h): the torch.distributed route, for backends without the library's RCCL path ----
class _DeviceBytes:
    """a raw device pointer as a CUDA-array-interface object (torch.as_tensor wraps it without copying)"""
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def exchange_light_vertices(renderer, rank, world_size):
    """every rank hands every other rank the light vertices of the batch in flight (fpt_bpt_export_light_vertices -> all_gather ->
    fpt_bpt_import_light_vertices); between GPUs use renderer.bpt_exchange_light_vertices() (RCCL inside the library) instead"""
    import torch
    import torch.distributed as dist
    ptr, n = renderer.bpt_export_light_vertices()
    mine = torch.as_tensor(_DeviceBytes(ptr, n * 80), device=renderer.dev).cpu().numpy().tobytes() if n else b""
    blobs = [None] * world_size
    dist.all_gather_object(blobs, mine)
    keep = []
    for j, blob in enumerate(blobs):
        if j != rank and blob:
            t = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(renderer.dev)
            torch.cuda.synchronize(renderer.dev)
            renderer.bpt_import_light_vertices(t.data_ptr(), len(blob) // 80)
            keep.append(t)
    renderer.synchronize()
