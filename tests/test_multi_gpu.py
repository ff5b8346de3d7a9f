"""GPU tests (-m gpu) of the N>1 path's RCCL legs: the C-ABI communicator entry points on one rank (runs on
the single-GPU box), and the 2-rank nccl run against the single-GPU frame when two GPUs are visible (skipped otherwise; the gloo
world_size-2 tests in test_distributed_cpu.py cover the sharding logic without GPUs)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SELFTEST = r"""
import ctypes as C, sys
sys.path.insert(0, %r)
import numpy as np, torch
import fermat_amd as fa
from fermat_amd import scene
from fermat_amd.distributed import comm_init, comm_info, gather_framebuffer_capi
s = scene.cornell_box("CornellBox-JP")
r = fa.Renderer(s, 64, 48, fa.default_options(4), gbuffer=False)
comm_init(r, 0, 1)                                           # fpt_comm_unique_id + fpt_comm_init: dlopen(librccl), ncclCommInitRank(1 rank)
assert comm_info(r) == (0, 1)                                # ncclCommUserRank / ncclCommCount
r._check(r.L.fpt_comm_selftest(r.ctx, C.c_uint32(100003)))     # grouped ncclSend + ncclRecv to self on the library's stream
r.render_pass(0)
before = r.framebuffer()[5].copy()
gather_framebuffer_capi(r, fa.tile_pixel_lists(64, 48, 1, tile=(64, 1)), root=0, channels=(5, 4))   # 1 rank: nothing travels, nothing changes
r.synchronize()
assert np.array_equal(before, r.framebuffer()[5])
assert r.L.fpt_comm_init(r.ctx, C.c_int(0), C.c_int(1), C.c_char_p(b"x" * 128)) != 0            # a second communicator is refused
r._check(r.L.fpt_comm_destroy(r.ctx))
assert r.L.fpt_gather_framebuffer(r.ctx, C.byref(r.view), C.c_int(0), C.c_uint32(32), None, None) != 0 and b"communicator" in r.L.fpt_last_error(r.ctx)
r.close()
# PSFPT under tile sharding, the RCCL route on a 1-rank communicator: all-reduce of the counts, merge of the own records, blend == the unsharded renderer
full = fa.Renderer(s, 64, 48, fa.default_options(5), psf_options=fa.default_psf_options())
p = fa.Renderer(s, 64, 48, fa.default_options(5), psf_options=fa.default_psf_options())
comm_init(p, 0, 1)
p.psf_set_sharded(True)
for i in range(3):
    full.psf_render(i, sync=True)
    p.psf_render(i); p.psf_exchange_cells(); p.psf_finish(sync=True)
a, b = full.psf_cells(), p.psf_cells()
assert np.array_equal(a["keys"], b["keys"]) and np.array_equal(a["counts"], b["counts"]) and np.array_equal(a["sums"], b["sums"])
assert np.array_equal(full.framebuffer().view(np.uint32), p.framebuffer().view(np.uint32))
full.close(); p.close()
print("RCCL_SELFTEST_OK")
""" % ROOT


def test_rccl_entry_points_on_one_rank():
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, "-c", SELFTEST], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0 and "RCCL_SELFTEST_OK" in p.stdout, (p.stdout[-1500:], p.stderr[-3000:])


@pytest.mark.parametrize("shape", ["1600x900 x3 ranks", "3840x2160 x8 ranks"])
def test_gather_halves_between_contexts_on_one_gpu(shape):
    """The N>1 gather without RCCL (VERDICT r3 task 2c): N rank contexts on ONE GPU render their scanline shares, every non-root packs its tiles
    (fpt_gather_pack), the message is moved with a device-to-device copy -- what RCCL's send / receive does between GPUs -- and the root scatters it
    (fpt_gather_unpack): the assembled frame is the single-context frame bit for bit, for BASELINE configs[2]'s and configs[3]'s frame sizes."""
    import numpy as np
    import torch
    import fermat_amd as fa
    from fermat_amd import scene
    from fermat_amd.distributed import set_tile_lists, gather_pack, gather_unpack
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    (W, H), world = ((1600, 900), 3) if shape.startswith("1600") else ((3840, 2160), 8)
    s = scene.cornell_box("CornellBox-Glossy")
    n, L = 2, 4
    lists = fa.tile_pixel_lists(W, H, world, tile=(W, 1))
    full = fa.Renderer(s, W, H, fa.default_options(L), gbuffer=False)
    full.set_batch(n); full.render_batch(0, n, sync=True)
    want = full.framebuffer()
    full.close()
    ranks = []
    for k in range(world):
        r = fa.Renderer(s, W, H, fa.default_options(L), pixels=lists[k], gbuffer=False)
        r.set_batch(n); r.render_batch(0, n)
        set_tile_lists(r, lists, k, root=0)
        ranks.append(r)
    root = ranks[0]
    for k in range(1, world):
        ptr, n_floats = gather_pack(ranks[k], channels=(5, 0))
        assert n_floats == len(lists[k]) * 4 * 2
        ranks[k].synchronize()
        msg = torch.empty(n_floats, dtype=torch.float32, device=root.dev)          # "the wire": a buffer the root owns
        assert hip.hipMemcpy(ctypes.c_void_p(msg.data_ptr()), ctypes.c_void_p(ptr), ctypes.c_size_t(n_floats * 4), ctypes.c_int(3)) == 0      # hipMemcpyDeviceToDevice
        assert hip.hipDeviceSynchronize() == 0          # a device-to-device hipMemcpy may return before it is done, and the library's stream does not wait for the null stream
        gather_unpack(root, k, msg.data_ptr(), channels=(5, 0))
        root.synchronize()
    got = root.framebuffer()
    for c in (5, 0):
        assert np.array_equal(got[c].view(np.uint32), want[c].view(np.uint32)), "channel %d of the assembled %s frame differs from the single-context frame" % (c, shape)
    # a channel that was not gathered still holds only the root's own scanlines
    own = np.zeros(W * H, bool); own[lists[0]] = True
    assert np.array_equal(got[1][own].view(np.uint32), want[1][own].view(np.uint32)) and not np.array_equal(got[1][~own].view(np.uint32), want[1][~own].view(np.uint32))
    # unpack is refused on a context whose tables name another root
    assert ranks[1].L.fpt_gather_unpack(ranks[1].ctx, ctypes.byref(ranks[1].view), ctypes.c_uint32(32), ctypes.c_int(0), ctypes.c_void_p(msg.data_ptr())) != 0
    for r in ranks:
        r.close()


def test_gather_call_costs_no_host_time_per_pixel():
    """fpt_gather_framebuffer no longer hashes or uploads the pixel lists per call (1.5 ms of host time per call at 1600x900, inside the timed
    region): with the tables registered, the call is a few microseconds of host time -- measured on a 1-rank communicator, where nothing travels."""
    import ctypes as C, time
    import numpy as np
    import fermat_amd as fa
    from fermat_amd import scene
    from fermat_amd.distributed import comm_init, set_tile_lists
    s = scene.cornell_box("CornellBox-JP")
    W, H = 3840, 2160
    r = fa.Renderer(s, W, H, fa.default_options(3), gbuffer=False)
    comm_init(r, 0, 1)
    lists = fa.tile_pixel_lists(W, H, 1, tile=(W, 1))
    set_tile_lists(r, lists, 0, root=0)
    arr = [np.ascontiguousarray(p, np.uint32) for p in lists]
    ptrs = (C.c_void_p * 1)(*[p.ctypes.data for p in arr]); counts = (C.c_uint32 * 1)(len(arr[0]))
    for form in ("registered", "passed again"):
        a, b = (None, None) if form == "registered" else (ptrs, counts)
        r._check(r.L.fpt_gather_framebuffer(r.ctx, C.byref(r.view), C.c_int(0), C.c_uint32(32), a, b))
        t0 = time.perf_counter()
        for _ in range(200):
            r._check(r.L.fpt_gather_framebuffer(r.ctx, C.byref(r.view), C.c_int(0), C.c_uint32(32), a, b))
        us = (time.perf_counter() - t0) / 200 * 1e6
        assert us < 50.0, "fpt_gather_framebuffer (%s tables, 8.3 M pixels): %.1f us of host time per call" % (form, us)
    r._check(r.L.fpt_comm_destroy(r.ctx))
    r.close()


def test_two_rank_rccl_gather_equals_single_gpu_frame():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the driver's 8-GPU box); one visible here")
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import socket
    with socket.socket() as sk:          # a free port: the 8-GPU box is shared
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "tests", "_multi_gpu_worker.py")], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0 and "MULTI_GPU_OK world=2" in p.stdout, (p.stdout[-1500:], p.stderr[-3000:])
    assert "RCCL_RANKS rank=0 ncclCommCount=2" in p.stdout and "RCCL_RANKS rank=1 ncclCommCount=2" in p.stdout, p.stdout[-1500:]
