"""world_size-2 `gloo` test of the N>1 path: tile sharding + frame-buffer gather.  The per-rank renderer is the
oracle here (CPU); the sharding/gather code is the product's (fermat_amd.api.tile_pixel_lists, fermat_amd.distributed)."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np, torch, torch.distributed as dist
    sys.path.insert(0, %r)
    from fermat_amd import scene
    from fermat_amd.api import tile_pixel_lists
    from fermat_amd.distributed import gather_framebuffer, gather_filter_inputs, allreduce_splats
    from oracle import binding as ob
    rank = int(os.environ["RANK"]); ws = int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    table = np.fromfile(os.path.join(scene.DATA_DIR, "glossy_reflectance.dat"), np.float32)
    s = scene.cornell_box()
    W, H = 40, 24
    lists = tile_pixel_lists(W, H, ws, tile=8)
    pt = ob.OraclePT(s, W, H, ob.default_options(4), table, scene.DATA_DIR)
    pt.clear_gbuffer()
    for i in range(2):
        pt.render_pass(i, lists[rank])
    out = gather_framebuffer(torch.from_numpy(pt.fb), lists, rank, ws, dst=0, channels=(5, 4))
    # kFiltered output under tile sharding: gather the filter's inputs + gbuffer, filter the assembled frame on rank 0
    fb_full, geo_full = gather_filter_inputs(torch.from_numpy(pt.fb), torch.from_numpy(pt.gb_geo), lists, rank, ws, dst=0)
    if rank == 0:
        np.save(os.environ["OUT"], out.numpy())
        pt.fb[...] = fb_full.numpy(); pt.gb_geo[...] = geo_full.numpy()
        pt.filter(1)
        np.save(os.environ["OUT"] + ".filtered.npy", pt.fb[6])
    # bidirectional path tracer: per-rank light + eye sub-paths, one integer all-reduce of the light-tracing splat sums per pass
    bp = ob.OraclePT(s, W, H, ob.default_options(4), table, scene.DATA_DIR)
    bp.bpt_init(ob.default_bpt_options(4), scene.DATA_DIR)
    splats = torch.from_numpy(bp.bpt_defer_splats())          # shares the oracle's buffer: the all-reduce lands in place
    for i in range(2):
        bp.bpt_render(i, lists[rank])
        allreduce_splats(splats, ws)
        bp.bpt_resolve_splats()
    outb = gather_framebuffer(torch.from_numpy(bp.fb), lists, rank, ws, dst=0, channels=(5, 4))
    if rank == 0:
        np.save(os.environ["OUT"] + ".bpt.npy", outb.numpy())
    dist.barrier()
    dist.destroy_process_group()
''') % ROOT


def test_two_rank_tile_render_and_gather(tmp_path, table, cornell):
    from fermat_amd import scene
    from oracle import binding as ob
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    script = tmp_path / "worker.py"; script.write_text(WORKER)
    out = tmp_path / "gathered.npy"
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OUT=str(out), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0
    gathered = np.load(out)
    full = ob.OraclePT(cornell, 40, 24, ob.default_options(4), table, scene.DATA_DIR)
    for i in range(2):
        full.render_pass(i)
    assert np.array_equal(gathered[0].view(np.uint32), full.fb[5].view(np.uint32))
    assert np.array_equal(gathered[1].view(np.uint32), full.fb[4].view(np.uint32))
    # the filtered image of the sharded run equals the single-process one
    full2 = ob.OraclePT(cornell, 40, 24, ob.default_options(4), table, scene.DATA_DIR)
    full2.clear_gbuffer()
    for i in range(2):
        full2.render_pass(i)
    full2.filter(1)
    fullb = ob.OraclePT(cornell, 40, 24, ob.default_options(4), table, scene.DATA_DIR)
    fullb.bpt_init(ob.default_bpt_options(4), scene.DATA_DIR)
    for i in range(2):
        fullb.bpt_render(i)
    gb = np.load(str(out) + ".bpt.npy")
    assert np.array_equal(gb[0].view(np.uint32), fullb.fb[5].view(np.uint32)) and np.array_equal(gb[1].view(np.uint32), fullb.fb[4].view(np.uint32))
    filtered = np.load(str(out) + ".filtered.npy")
    assert np.array_equal(filtered.view(np.uint32), full2.fb[6].view(np.uint32))
