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
