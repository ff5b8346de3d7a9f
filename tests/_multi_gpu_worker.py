"""worker of tests/test_multi_gpu.py (run under torch.distributed.run, one rank per GPU, backend nccl = RCCL): renders its scanline share,
gathers COMPOSITED_C to rank 0 through the C-ABI (fpt_gather_framebuffer) AND through torch.distributed, and rank 0 checks both against
the single-GPU frame bit for bit."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    import fermat_amd as fa
    from fermat_amd import scene
    from fermat_amd.distributed import comm_init, comm_info, gather_framebuffer, gather_framebuffer_capi, set_tile_lists
    rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"]); local = int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    W, H, L, n = 256, 128, 5, 3
    s = scene.cornell_box("CornellBox-Glossy")
    lists = fa.tile_pixel_lists(W, H, world, tile=(W, 1))
    r = fa.Renderer(s, W, H, fa.default_options(L), device=local, pixels=lists[rank], gbuffer=False)
    r.set_batch(n)
    r.render_batch(0, n, sync=True)
    via_torch = gather_framebuffer(r.fb, lists, rank, world, dst=0, channels=(5,))
    comm_init(r, rank, world)
    got_rank, got_world = comm_info(r)          # what RCCL itself says (ncclCommUserRank / ncclCommCount): no silent 1-rank fallback
    assert (got_rank, got_world) == (rank, world), "the library's RCCL communicator reports rank %d of %d, expected %d of %d" % (got_rank, got_world, rank, world)
    print("RCCL_RANKS rank=%d ncclCommCount=%d" % (got_rank, got_world), flush=True)
    set_tile_lists(r, lists, rank, root=0)
    gather_framebuffer_capi(r, None, root=0, channels=(5,))          # the registered tables, stream-ordered behind the render: no synchronize in front
    r.synchronize()
    if rank == 0:
        full = fa.Renderer(s, W, H, fa.default_options(L), device=local, gbuffer=False)
        full.set_batch(n)
        full.render_batch(0, n, sync=True)
        want = full.framebuffer()[5]
        got = r.framebuffer()[5]
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "fpt_gather_framebuffer: the gathered frame differs from the single-GPU frame"
        assert np.array_equal(via_torch[0].cpu().numpy().view(np.uint32), want.view(np.uint32)), "torch.distributed gather differs"
    dist.barrier()
    r.close()
    # PSFPT: the ranks exchange their cache cells after every pass (fpt_psfpt_exchange_cells over RCCL); rank 0 compares its scanlines and its
    # copy of the global table with the single-GPU renderer's
    p = fa.Renderer(s, W, H, fa.default_options(L), device=local, pixels=lists[rank], psf_options=fa.default_psf_options(psf_temporal_reuse=2))
    comm_init(p, rank, world)
    p.psf_set_sharded(True)
    for i in range(3):
        p.psf_render(i); p.psf_exchange_cells(); p.psf_finish(sync=True)
    if rank == 0:
        full = fa.Renderer(s, W, H, fa.default_options(L), device=local, psf_options=fa.default_psf_options(psf_temporal_reuse=2))
        for i in range(3):
            full.psf_render(i, sync=True)
        a, b = full.psf_cells(), p.psf_cells()
        assert np.array_equal(a["keys"], b["keys"]) and np.array_equal(a["counts"], b["counts"]) and np.array_equal(a["sums"], b["sums"]), "PSFPT: the merged cache differs"
        want, got = full.framebuffer(), p.framebuffer()
        for c in range(8):
            assert np.array_equal(got[c][lists[0]].view(np.uint32), want[c][lists[0]].view(np.uint32)), "PSFPT: channel %d of rank 0's scanlines differs" % c
        full.close()
        print("MULTI_GPU_OK world=%d" % world)
    dist.barrier()
    p.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
