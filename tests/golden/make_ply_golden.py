#!/usr/bin/env python3
"""Regenerates tests/golden/ply_cases.npz: what the reference's own PLY parser (rply 1.01, driven like MeshBase::loadFromPly by
oracle/_ref/ply_dump -- `make -C oracle ref`, needs /root/reference) returns for the files of tests/ply_cases.py.
Run from the repo root in the build container:  python tests/golden/make_ply_golden.py"""
import os, subprocess, sys, tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ply_cases


def ref_dump(path):
    exe = os.path.join(ROOT, "oracle", "_ref", "ply_dump")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "o.bin")
        if subprocess.call([exe, path, out]) != 0:
            return None
        b = open(out, "rb").read()
    nv, nn, nt, ntri = np.frombuffer(b, np.int32, 4)
    o = 16
    P = np.frombuffer(b, np.float32, nv * 3, o).reshape(-1, 3); o += nv * 12
    N = np.frombuffer(b, np.float32, nn * 3, o).reshape(-1, 3); o += nn * 12
    T = np.frombuffer(b, np.float32, nt * 2, o).reshape(-1, 2); o += nt * 8
    tri = np.frombuffer(b, np.int32, ntri * 3, o).reshape(-1, 3)
    return dict(P=P.copy(), N=N.copy(), T=T.copy(), tri=tri.copy())


if __name__ == "__main__":
    with tempfile.TemporaryDirectory() as d:
        files = ply_cases.cases(d)
        out = {}
        for name, path in files.items():
            r = ref_dump(path)
            assert r is not None, name
            for k, v in r.items():
                out[name + "." + k] = v
            print(name, {k: v.shape for k, v in r.items()})
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ply_cases.npz"), **out)
