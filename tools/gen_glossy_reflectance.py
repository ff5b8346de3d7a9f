#!/usr/bin/env python3
"""Regenerate fermat_amd/data/glossy_reflectance.dat (32^4 fp32, 4 MiB).

The reference hard-requires vs/fermat/glossy_reflectance.dat (src/renderer.cu:646-660) but the blob is missing from the
checkout (.MISSING_LARGE_BLOBS).  This script restates its generator (src/bsdf.cu:36-102, all S^4 cells, SURVEY
Appendix E) through the oracle's C restatement and writes the table.  The table is an INPUT FIXTURE of both the oracle
and the HIP product (like a texture); it cannot be proven equal to NVIDIA's shipped file.  ~12 min on 8 cores.
"""
import ctypes, hashlib, os, sys
import numpy as np

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(root, "oracle", "liboracle.so"))
S = 32
N = S ** 4
out = np.zeros(N, dtype=np.float32)
step = 1 << 15
for b in range(0, N, step):
    e = min(N, b + step)
    lib.orc_glossy_reflectance_cells(ctypes.c_uint32(b), ctypes.c_uint32(e), ctypes.c_void_p(out[b:e].ctypes.data))
    print("\r%5.1f%%" % (100.0 * e / N), end="", file=sys.stderr)
dst = os.path.join(root, "fermat_amd", "data", "glossy_reflectance.dat")
out.tofile(dst)
print("\nwrote", dst, "sha256", hashlib.sha256(out.tobytes()).hexdigest())
