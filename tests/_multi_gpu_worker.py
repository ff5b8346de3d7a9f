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
    from fermat_amd.distributed import comm_init, gather_framebuffer, gather_framebuffer_capi
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
    gather_framebuffer_capi(r, lists, root=0, channels=(5,))
    r.synchronize()
    if rank == 0:
        full = fa.Renderer(s, W, H, fa.default_options(L), device=local, gbuffer=False)
        full.set_batch(n)
        full.render_batch(0, n, sync=True)
        want = full.framebuffer()[5]
        got = r.framebuffer()[5]
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "fpt_gather_framebuffer: the gathered frame differs from the single-GPU frame"
        assert np.array_equal(via_torch[0].cpu().numpy().view(np.uint32), want.view(np.uint32)), "torch.distributed gather differs"
        print("MULTI_GPU_OK world=%d" % world)
    dist.barrier()
    r.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
