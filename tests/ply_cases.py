"""Deterministic PLY test files (ascii / little / big endian, extra properties, quads, short faces, comments, extra elements)
shared by tests/test_ply.py and tests/golden/make_ply_golden.py.  The files are OUR data, written here byte by byte."""
import struct

import numpy as np

_FMT = {"char": "b", "uchar": "B", "short": "h", "ushort": "H", "int": "i", "uint": "I", "float": "f", "double": "d",
        "int8": "b", "uint8": "B", "int16": "h", "uint16": "H", "int32": "i", "uint32": "I", "float32": "f", "float64": "d"}


def _emit(mode, typ, v):
    if mode == "ascii":
        return (repr(float(v)) if _FMT[typ] in "fd" else str(int(v))).encode() + b" "
    return struct.pack(("<" if mode == "binary_little_endian" else ">") + _FMT[typ], v if _FMT[typ] in "fd" else int(v))


def write_ply(path, mode, elements, comments=(), eol=b"\n"):
    """elements: [(name, [(prop, type) | (prop, ("list", len type, value type))], rows)] with rows = list of per-instance value lists"""
    out = [b"ply\n", b"format " + mode.encode() + b" 1.0" + eol]
    for c in comments:
        out.append(c.encode() + eol)
    for name, props, rows in elements:
        out.append(("element %s %d" % (name, len(rows))).encode() + eol)
        for pn, t in props:
            if isinstance(t, tuple):
                out.append(("property list %s %s %s" % (t[1], t[2], pn)).encode() + eol)
            else:
                out.append(("property %s %s" % (t, pn)).encode() + eol)
    out.append(b"end_header\n")
    for name, props, rows in elements:
        for row in rows:
            for (pn, t), v in zip(props, row):
                if isinstance(t, tuple):
                    out.append(_emit(mode, t[1], len(v)))
                    for x in v:
                        out.append(_emit(mode, t[2], x))
                else:
                    out.append(_emit(mode, t, v))
            if mode == "ascii":
                out.append(b"\n")
    open(path, "wb").write(b"".join(out))


def _grid(rng, n):
    """n x n vertex grid with normals and uvs, two triangles per cell"""
    xs, ys = np.meshgrid(np.linspace(-1, 1, n), np.linspace(-1, 1, n))
    P = np.stack([xs.ravel(), ys.ravel(), 0.3 * rng.standard_normal(n * n)], 1).astype(np.float32)
    N = rng.standard_normal((n * n, 3)).astype(np.float32)
    N /= np.linalg.norm(N, axis=1, keepdims=True)
    T = (np.stack([xs.ravel(), ys.ravel()], 1) * 0.5 + 0.5).astype(np.float32)
    F = []
    for j in range(n - 1):
        for i in range(n - 1):
            a = j * n + i
            F.append([a, a + 1, a + n + 1]); F.append([a, a + n + 1, a + n])
    return P, N, T, F


def cases(tmp):
    """writes the case files under `tmp`; returns {case name: path}"""
    import os
    rng = np.random.RandomState(7)
    out = {}
    P, N, T, F = _grid(rng, 5)
    # a: ascii, positions + normals + s/t + colours (ignored), triangles
    rows = [list(map(float, P[i])) + list(map(float, N[i])) + list(map(float, T[i])) + [int(v) for v in rng.randint(0, 256, 3)] for i in range(len(P))]
    out["ascii_full"] = os.path.join(tmp, "a.ply")
    write_ply(out["ascii_full"], "ascii",
              [("vertex", [("x", "float"), ("y", "float"), ("z", "float"), ("nx", "float"), ("ny", "float"), ("nz", "float"),
                           ("s", "float"), ("t", "float"), ("red", "uchar"), ("green", "uchar"), ("blue", "uchar")], rows),
               ("face", [("vertex_indices", ("list", "uchar", "int"))], [[f] for f in F])],
              comments=["comment made by tests/ply_cases.py", "obj_info nothing to see"])
    # b: little endian, double positions, u/v, quads (4th index dropped), a list property before vertex_indices, an extra element in between
    quads = [[f[0], f[1], f[2], (f[2] + 1) % len(P)] for f in F[:9]]
    out["le_double_quads"] = os.path.join(tmp, "b.ply")
    write_ply(out["le_double_quads"], "binary_little_endian",
              [("vertex", [("x", "double"), ("y", "double"), ("z", "double"), ("u", "float32"), ("v", "float32")],
                [list(map(float, P[i])) + list(map(float, T[i])) for i in range(len(P))]),
               ("edge", [("a", "int32"), ("b", "int32"), ("w", "float64")], [[i, i + 1, 0.5 * i] for i in range(7)]),
               ("face", [("weights", ("list", "uint8", "float32")), ("vertex_indices", ("list", "uint8", "uint32")), ("flags", "ushort")],
                [[[0.25, 0.75], q, 3] for q in quads])],
              comments=["comment binary case"])
    # c: big endian, normals but no texture coordinates, int16 list length, faces of 2 (overwritten), 3 and 5 entries
    faces = [[F[0][0], F[0][1]], F[1], F[2] + [F[2][0], F[2][1]], F[3]]
    out["be_short_faces"] = os.path.join(tmp, "c.ply")
    write_ply(out["be_short_faces"], "binary_big_endian",
              [("vertex", [("x", "float"), ("y", "float"), ("z", "float"), ("nx", "float"), ("ny", "float"), ("nz", "float")],
                [list(map(float, P[i])) + list(map(float, N[i])) for i in range(len(P))]),
               ("face", [("vertex_indices", ("list", "int16", "int"))], [[f] for f in faces])])
    # d: ascii, positions only, face list called vertex_index (the loader looks for vertex_indices only -> no triangles)
    out["ascii_vertex_index"] = os.path.join(tmp, "d.ply")
    write_ply(out["ascii_vertex_index"], "ascii",
              [("vertex", [("x", "float"), ("y", "float"), ("z", "float")], [list(map(float, P[i])) for i in range(6)]),
               ("face", [("vertex_index", ("list", "uchar", "int"))], [[f] for f in F[:3]])])
    # e: ascii with y before x and both s/t ... u/v absent, integers given for float properties, CRLF header lines
    out["ascii_yxz_crlf"] = os.path.join(tmp, "e.ply")
    write_ply(out["ascii_yxz_crlf"], "ascii",
              [("vertex", [("y", "float"), ("x", "float"), ("z", "float"), ("s", "double"), ("t", "double")],
                [[float(P[i][1]), float(P[i][0]), float(int(10 * P[i][2])), float(T[i][0]), float(T[i][1])] for i in range(len(P))]),
               ("face", [("vertex_indices", ("list", "uint", "uint"))], [[f] for f in F[:5]])], eol=b"\r\n")
    return out
