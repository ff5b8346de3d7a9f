#!/usr/bin/env python3
"""Writes fermat_amd/data/scenes/bathroom2_standin/: the stand-in for BASELINE configs[2]-[3]'s scene, models/bathroom2 (VERDICT r3 task 5).

What the reference checkout HAS of bathroom2 is used as it is: bathroom.mtl (23 materials: Kd + Ks maps, mirror, lacquer, ceramic, emitters), its 28 texture
files (128 x 128 TGA) and the camera of bathroom.fa -- copied as data with their licence (CC-BY, 'Salle de bain' by nacimus via B. Bitterli's resources).
What it LACKS is bathroom.obj (.MISSING_LARGE_BLOBS): the geometry here is procedural -- a room of bathroom2's extent seen from bathroom2's own camera with
a bathtub, a wash-stand with basin, tap and mirror, a toilet, shelves full of bottles, towels on a rail, window blinds in front of a sky light, tiled
walls, floor planks, ceiling lights -- instanced from a dozen small PLY parts through a .fa script (src/mesh/fermat_loader.cpp's format), so that the scene goes
through the same front-end a Fermat user's scene would and every surface wears one of bathroom.mtl's materials.  Unlike round 1-3's stand-in (six big spheres
in an open box: 3.3 node steps per ray) the room is full of medium and small objects at every depth.

  python tools/gen_bathroom2_standin.py          # deterministic: rewrites the committed files bit for bit
"""
import os
import shutil
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "fermat_amd", "data", "scenes", "bathroom2_standin")
REF = "/root/reference/models/bathroom2"


# ---- part meshes ------------------------------------------------------------------------------------------------------------------
def write_ply(path, pos, nrm, uv, tri):
    """binary little-endian PLY with x y z nx ny nz s t per vertex (what MeshBase::loadFromPly reads through rply)"""
    n, m = len(pos), len(tri)
    hdr = ("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
           "property float nx\nproperty float ny\nproperty float nz\nproperty float s\nproperty float t\n"
           "element face %d\nproperty list uchar int vertex_indices\nend_header\n" % (n, m)).encode()
    v = np.concatenate([pos, nrm, uv], 1).astype("<f4")
    f = np.empty(m, dtype=[("n", "u1"), ("i", "<i4", 3)])
    f["n"] = 3; f["i"] = tri
    with open(path, "wb") as fh:
        fh.write(hdr); fh.write(v.tobytes()); fh.write(f.tobytes())


def grid_mesh(P, nu, nv, flip=False):
    """triangulate an (nu+1) x (nv+1) lattice of points P[u, v] (with normals from the lattice, uv = lattice coordinates)"""
    idx = np.arange((nu + 1) * (nv + 1)).reshape(nu + 1, nv + 1)
    a, b, c, d = idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel(), idx[:-1, 1:].ravel()
    tri = np.concatenate([np.stack([a, b, c], 1), np.stack([a, c, d], 1)])
    if flip:
        tri = tri[:, ::-1]
    du = np.gradient(P, axis=0); dv = np.gradient(P, axis=1)
    n = np.cross(du, dv); n /= np.maximum(np.linalg.norm(n, axis=2, keepdims=True), 1e-20)
    if flip:
        n = -n
    return P.reshape(-1, 3), n.reshape(-1, 3), tri


def join(parts):
    pos, nrm, uv, tri, base = [], [], [], [], 0
    for p, n, t, f in parts:
        pos.append(p); nrm.append(n); uv.append(t); tri.append(f + base); base += len(p)
    return np.concatenate(pos), np.concatenate(nrm), np.concatenate(uv), np.concatenate(tri)


def lattice_uv(nu, nv, su=1.0, sv=1.0):
    u, v = np.meshgrid(np.linspace(0, su, nu + 1), np.linspace(0, sv, nv + 1), indexing="ij")
    return np.stack([u, v], 2).reshape(-1, 2)


def part_quad(n=8, rep=4.0):
    """unit square in the xz plane, y up, centred, n x n cells, texture repeated rep times"""
    u, v = np.meshgrid(np.linspace(-0.5, 0.5, n + 1), np.linspace(-0.5, 0.5, n + 1), indexing="ij")
    P = np.stack([u, np.zeros_like(u), -v], 2)
    p, nr, t = grid_mesh(P, n, n)
    return p, nr, lattice_uv(n, n, rep, rep), t


def part_box(bevel=0.06):
    """unit cube centred at the origin with bevelled edges: a superquadric x^8 + y^8 + z^8 = 1 sampled on a 12 x 24 lattice"""
    nu, nv = 12, 24
    th = np.linspace(-np.pi / 2, np.pi / 2, nu + 1); ph = np.linspace(-np.pi, np.pi, nv + 1)
    T, Ph = np.meshgrid(th, ph, indexing="ij")
    e = 0.25
    f = lambda w, m: np.sign(w) * np.abs(w) ** m   # noqa: E731
    P = 0.5 * np.stack([f(np.cos(T), e) * f(np.cos(Ph), e), f(np.sin(T), e), f(np.cos(T), e) * f(np.sin(Ph), e)], 2)
    p, n, t = grid_mesh(P, nu, nv, flip=True)
    return p, n, lattice_uv(nu, nv, 1.0, 2.0), t


def part_cylinder(seg=32, rings=6):
    """unit cylinder (radius 0.5, height 1, y axis) with rounded caps"""
    prof_r = np.concatenate([np.linspace(0, 0.5, 4)[:-1], np.full(rings + 1, 0.5), np.linspace(0.5, 0, 4)[1:]])
    prof_y = np.concatenate([np.full(3, -0.5), np.linspace(-0.5, 0.5, rings + 1), np.full(3, 0.5)])
    return revolve(prof_r, prof_y, seg)


def revolve(prof_r, prof_y, seg, rep_u=2.0, rep_v=1.0):
    a = np.linspace(0, 2 * np.pi, seg + 1)
    R, A = np.meshgrid(prof_r, a, indexing="ij"); Y, _ = np.meshgrid(prof_y, a, indexing="ij")
    P = np.stack([R * np.cos(A), Y, R * np.sin(A)], 2)
    p, n, t = grid_mesh(P, len(prof_r) - 1, seg, flip=True)
    return p, n, lattice_uv(len(prof_r) - 1, seg, rep_v, rep_u), t


def part_sphere(n_lat=48):
    th = np.linspace(-np.pi / 2, np.pi / 2, n_lat + 1)
    return revolve(0.5 * np.cos(th), 0.5 * np.sin(th), 2 * n_lat)


def part_bottle(seg=64):
    """a bottle: body, shoulder, neck, cap -- surface of revolution, height 1, widest radius 0.22"""
    y = np.linspace(0, 1, 97)
    r = np.where(y < 0.55, 0.22, np.where(y < 0.75, 0.22 - 0.14 * (np.clip(y - 0.55, 0.0, 1.0) / 0.2) ** 1.5, np.where(y < 0.9, 0.08, 0.1)))
    r = r * np.minimum(1.0, np.minimum(y / 0.03 + 0.15, (1.0 - y) / 0.02 + 0.05))
    r[0] = 0.0; r[-1] = 0.0
    return revolve(r, y, seg)


def part_bowl(n=128, wall=0.06):
    """a hollow half-ellipsoid shell, open at the top (rim at y = 0, bottom at y = -0.5), unit footprint: bathtub, basin, toilet bowl"""
    th = np.linspace(0, np.pi / 2, n // 2 + 1)
    outer = revolve(0.5 * np.cos(th)[::-1] ** 0.6, -0.5 * np.sin(th)[::-1], n)
    ri = (0.5 - wall) * np.cos(th) ** 0.6
    inner = revolve(ri, -(0.5 - wall) * np.sin(th), n)
    rim = revolve(np.linspace(0.5 - wall, 0.5, 4), np.zeros(4), n)
    return join([outer, inner, rim])


def part_torus(nu=48, nv=96, r=0.12):
    a = np.linspace(0, 2 * np.pi, nu + 1); b = np.linspace(0, 2 * np.pi, nv + 1)
    A, B = np.meshgrid(a, b, indexing="ij")
    P = np.stack([(0.5 + r * np.cos(A)) * np.cos(B), r * np.sin(A), (0.5 + r * np.cos(A)) * np.sin(B)], 2)
    p, n, t = grid_mesh(P, nu, nv, flip=True)
    return p, n, lattice_uv(nu, nv, 1.0, 4.0), t


def part_cloth(n=192, seed=3):
    """a towel hanging over a rail: a unit sheet folded over y = 0 (both halves hang down), with folds and a little noise"""
    rng = np.random.RandomState(seed)
    u, v = np.meshgrid(np.linspace(0, 1, n + 1), np.linspace(-1, 1, n + 1), indexing="ij")
    ph = rng.uniform(0, 6.28, 4)
    fold = 0.035 * np.sin(14 * u + ph[0]) + 0.02 * np.sin(31 * u + 3 * v + ph[1]) + 0.012 * np.sin(55 * u - 7 * v + ph[2])
    drape = np.abs(v)
    x = u - 0.5 + 0.01 * np.sin(9 * v + ph[3])
    y = -drape * 0.9 + 0.02 * np.cos(14 * u + ph[0]) * drape
    z = np.sign(v) * (0.04 + 0.03 * np.sqrt(drape)) + fold * (0.3 + drape)
    z = np.where(np.abs(v) < 0.06, v / 0.06 * 0.045, z)
    y = np.where(np.abs(v) < 0.06, 0.045 * np.cos(v / 0.06 * np.pi / 2) - 0.045 + y, y)
    P = np.stack([x, y, z], 2)
    p, nr, t = grid_mesh(P, n, n)
    return p, nr, lattice_uv(n, n, 3.0, 6.0), t


PARTS = {"quad": part_quad, "box": part_box, "cylinder": part_cylinder, "sphere": part_sphere, "bottle": part_bottle, "bowl": part_bowl,
         "torus": part_torus, "cloth": part_cloth}


# ---- the scene script ---------------------------------------------------------------------------------------------------------------
class Script:
    def __init__(self):
        self.lines = []; self.count = 0; self.tris = 0

    def add(self, part, mat, scale=(1, 1, 1), rot=(0, 0, 0), at=(0, 0, 0)):
        """instance: Scale, then RotateX / RotateZ / RotateY (degrees), then Translate -- the .fa loader applies the commands of a block in the order they are written"""
        L = ["Begin"]
        L.append("\tScale %.4f %.4f %.4f" % tuple(scale))
        if rot[0]: L.append("\tRotateX %.3f" % rot[0])
        if rot[2]: L.append("\tRotateZ %.3f" % rot[2])
        if rot[1]: L.append("\tRotateY %.3f" % rot[1])
        L.append("\tTranslate %.4f %.4f %.4f" % tuple(at))
        L.append("\tSetMaterial %s" % mat)
        L.append("\tLoadMesh %s.ply" % part)
        L.append("End")
        self.lines.append("\n".join(L)); self.count += 1; self.tris += TRI_COUNT[part]


TRI_COUNT = {}


def build_script():
    S = Script()
    rng = np.random.RandomState(2027)
    X0, X1, Y1, Z0, Z1 = -22.0, 18.0, 30.0, -24.0, 36.0          # the room; the camera of bathroom.fa stands at z = 32.3 and looks down -z
    cx, cz = 0.5 * (X0 + X1), 0.5 * (Z0 + Z1)
    W, D = X1 - X0, Z1 - Z0
    # shell: floor boards, walls, ceiling
    S.add("quad", "Wall", (W, 1, D), (180, 0, 0), (cx, Y1, cz))                     # ceiling (facing down)
    S.add("quad", "Wall", (W, 1, Y1), (90, 0, 0), (cx, Y1 / 2, Z0))                 # back wall
    S.add("quad", "Wall", (W, 1, Y1), (-90, 0, 0), (cx, Y1 / 2, Z1))                # wall behind the camera
    S.add("quad", "Wall", (Y1, 1, D), (0, 0, -90), (X0, Y1 / 2, cz))                # left wall
    S.add("quad", "Wall", (Y1, 1, D), (0, 0, 90), (X1, Y1 / 2, cz))                 # right wall
    S.add("quad", "BlackWall", (W, 1, D), (0, 0, 0), (cx, -0.05, cz))               # sub-floor
    n_planks = 26
    for k in range(n_planks):                                                        # floor boards, individually bevelled
        w = W / n_planks
        S.add("box", "WoodFloor", (w * 0.985, 0.5, D), (0, 0, 0), (X0 + (k + 0.5) * w, 0.2, cz))
    for z, y in ((Z0 + 0.3, 0.9), (Z1 - 0.3, 0.9)):                                  # skirting
        S.add("box", "Trims", (W, 1.8, 0.5), (0, 0, 0), (cx, y, z))
    for x in (X0 + 0.3, X1 - 0.3):
        S.add("box", "Trims", (0.5, 1.8, D), (0, 0, 0), (x, 0.9, cz))
    # tiled band on the back wall and on the right wall behind the bathtub (bevelled tiles, a dark border row on top)
    for k in range(20):
        for j in range(6):
            S.add("box", "Ceramic" if (j < 5) else "DarkBorder", (1.93, 1.93, 0.35), (0, 0, 0), (X0 + 1.0 + 2.0 * k, 2.9 + 2.0 * j, Z0 + 0.2))
    for k in range(16):
        for j in range(6):
            S.add("box", "Ceramic" if (j < 5) else "DarkBorder", (0.35, 1.93, 1.93), (0, 0, 0), (X1 - 0.2, 2.9 + 2.0 * j, -22.0 + 2.0 * k))
    # ceiling lights: four recessed panels (the Light material is the scene's emitter) with lacquered frames
    for lx, lz in ((-10, -8), (6, -8), (-10, 14), (6, 14)):
        S.add("quad", "Light", (5.0, 1, 5.0), (180, 0, 0), (lx, Y1 - 0.35, lz))
        for dx, dz, sx, sz in ((0, 2.7, 5.8, 0.4), (0, -2.7, 5.8, 0.4), (2.7, 0, 0.4, 5.8), (-2.7, 0, 0.4, 5.8)):
            S.add("box", "BlackWoodLacquer", (sx, 0.7, sz), (0, 0, 0), (lx + dx, Y1 - 0.35, lz + dz))
    # window in the left wall: a sky-light panel behind blinds
    S.add("quad", "SkyLight", (10.0, 1, 14.0), (0, 0, -90), (X0 + 0.15, 17.0, 6.0))
    for k in range(28):
        S.add("box", "Plastic", (1.6, 0.08, 14.0), (0, 0, 32.0), (X0 + 1.1, 12.3 + 0.36 * k, 6.0))
    for z in (-1.3, 13.3):
        S.add("box", "Wood", (0.9, 11.0, 0.7), (0, 0, 0), (X0 + 0.5, 17.0, z))
    for y in (11.6, 22.4):
        S.add("box", "Wood", (0.9, 0.7, 15.3), (0, 0, 0), (X0 + 0.5, y, 6.0))
    # wash-stand on the back wall: cabinet, top, basin, tap, mirror with frame
    S.add("box", "Wood", (14.0, 7.0, 5.0), (0, 0, 0), (-2.0, 3.9, Z0 + 2.9))
    S.add("box", "BlackWoodLacquer", (15.0, 0.7, 6.0), (0, 0, 0), (-2.0, 7.75, Z0 + 3.2))
    S.add("bowl", "Ceramic", (7.0, 3.2, 4.4), (0, 0, 0), (-2.0, 9.75, Z0 + 3.4))
    for k in range(4):
        S.add("box", "Wood", (3.2, 5.6, 0.25), (0, 0, 0), (-7.1 + 3.4 * k, 3.9, Z0 + 5.5))                    # doors
        S.add("sphere", "StainlessRough", (0.4, 0.4, 0.4), (0, 0, 0), (-6.0 + 3.4 * k, 4.4, Z0 + 5.8))       # knobs
    S.add("cylinder", "StainlessRough", (0.5, 2.6, 0.5), (0, 0, 0), (-2.0, 11.0, Z0 + 1.3))
    S.add("torus", "StainlessRough", (2.2, 2.2, 2.2), (90, 0, 0), (-2.0, 12.2, Z0 + 2.4))
    S.add("sphere", "StainlessRough", (0.9, 0.9, 0.9), (0, 0, 0), (-3.6, 10.2, Z0 + 1.4))
    S.add("sphere", "StainlessRough", (0.9, 0.9, 0.9), (0, 0, 0), (-0.4, 10.2, Z0 + 1.4))
    S.add("quad", "Mirror", (12.0, 1, 9.0), (90, 0, 0), (-2.0, 18.0, Z0 + 0.45))
    for dx, dy, sx, sy in ((0, 4.8, 13.2, 0.6), (0, -4.8, 13.2, 0.6), (6.3, 0, 0.6, 10.2), (-6.3, 0, 0.6, 10.2)):
        S.add("box", "BlackWoodLacquer", (sx, sy, 0.6), (0, 0, 0), (-2.0 + dx, 18.0 + dy, Z0 + 0.6))
    # shelves left and right of the mirror (where bathroom_cornell.fa puts its boxes), full of bottles and jars
    for sx in (-14.5, 10.5):
        for sy in (9.0, 13.0, 17.0, 21.0, 25.0):
            S.add("box", "Wood", (6.0, 0.35, 3.4), (0, 0, 0), (sx, sy, Z0 + 2.0))
            for k in range(6):
                h = rng.uniform(1.6, 3.2); r = rng.uniform(1.4, 2.6)
                mat = ("Plastic", "DarkPlastic", "Label", "Bin")[rng.randint(4)]
                if rng.rand() < 0.75:
                    S.add("bottle", mat, (r, h, r), (0, rng.uniform(0, 360), 0), (sx - 2.4 + 0.95 * k + rng.uniform(-0.1, 0.1), sy + 0.18, Z0 + 1.4 + rng.uniform(0, 1.2)))
                else:
                    S.add("sphere", mat, (1.1, 1.1, 1.1), (0, 0, 0), (sx - 2.4 + 0.95 * k, sy + 0.73, Z0 + 1.8))
        for side in (-3.0, 3.0):
            S.add("box", "Wood", (0.35, 18.5, 3.4), (0, 0, 0), (sx + side, 16.2, Z0 + 2.0))
    # bathtub along the right wall, with taps, bottles on the rim and a bath mat in front
    S.add("bowl", "Ceramic", (11.0, 9.5, 26.0), (0, 0, 0), (11.5, 9.6, -8.0))
    S.add("box", "Ceramic", (11.6, 0.5, 26.6), (0, 0, 0), (11.5, 0.45, -8.0))
    S.add("cylinder", "StainlessRough", (0.5, 3.0, 0.5), (0, 0, 0), (11.5, 11.0, -20.3))
    S.add("torus", "StainlessRough", (2.6, 2.6, 2.6), (90, 0, 0), (11.5, 12.4, -19.0))
    for k in range(9):
        h = rng.uniform(1.8, 3.4); r = rng.uniform(1.5, 2.4)
        S.add("bottle", ("Plastic", "DarkPlastic", "Label")[k % 3], (r, h, r), (0, rng.uniform(0, 360), 0), (16.6 + rng.uniform(-0.2, 0.2), 9.65, -19.0 + 2.4 * k))
    S.add("cloth", "Towel", (9.0, 0.02, 12.0), (90, 0, 0), (3.5, 0.75, -8.0))                                       # bath mat: the sheet laid flat
    # toilet and bin on the left
    S.add("bowl", "Ceramic", (5.0, 5.5, 6.5), (0, 0, 0), (-16.5, 5.8, 22.0))
    S.add("torus", "Ceramic", (5.0, 2.0, 6.5), (0, 0, 0), (-16.5, 5.9, 22.0))
    S.add("box", "Ceramic", (6.0, 7.0, 2.4), (0, 0, 0), (-16.5, 8.0, 26.6))
    S.add("cylinder", "Bin", (3.2, 4.4, 3.2), (0, 0, 0), (-10.5, 2.6, 27.5))
    S.add("torus", "StainlessRough", (3.3, 1.0, 3.3), (0, 0, 0), (-10.5, 4.8, 27.5))
    # towel rails on the left wall and a free-standing towel ladder in the room, with hanging towels
    for k, (tx, ty, tz, ry) in enumerate(((X0 + 1.6, 16.0, -14.0, 90), (X0 + 1.6, 16.0, -6.0, 90), (-6.5, 15.0, 9.0, 20), (-6.1, 10.5, 9.6, 20), (6.5, 13.0, 24.0, -35))):
        S.add("cylinder", "StainlessRough", (0.3, 7.4, 0.3), (0, ry, 90), (tx, ty, tz))
        S.add("cloth", "Towel", (6.4, 7.5, 6.0), (0, ry, 0), (tx, ty + 0.15, tz))
    for lx, lz in ((-9.9, 7.8), (-3.0, 10.3)):
        S.add("cylinder", "Wood", (0.45, 17.0, 0.45), (0, 0, 0), (lx, 8.7, lz))
    S.add("cylinder", "StainlessRough", (0.35, 17.0, 0.35), (0, 0, 0), (6.5, 8.7, 24.0))
    S.add("cylinder", "BlackWoodLacquer", (3.0, 0.5, 3.0), (0, 0, 0), (6.5, 0.7, 24.0))
    # a stool with a stack of folded towels and a basket of balls in the middle of the room
    S.add("cylinder", "Wood", (5.0, 0.6, 5.0), (0, 0, 0), (-1.5, 5.2, 17.0))
    for a in range(4):
        S.add("cylinder", "Wood", (0.5, 5.0, 0.5), (0, 0, 0), (-1.5 + 1.8 * np.cos(a * 1.5708 + 0.6), 2.6, 17.0 + 1.8 * np.sin(a * 1.5708 + 0.6)))
    for k in range(5):
        S.add("box", "Towel", (3.6, 0.55, 2.8), (0, 12.0 * k, 0), (-1.5, 5.8 + 0.58 * k, 17.0))
    S.add("bowl", "Bin", (6.0, 3.6, 6.0), (0, 0, 0), (5.0, 4.1, 8.0))
    for k in range(14):
        a = rng.uniform(0, 6.28); rr = rng.uniform(0, 1.7)
        S.add("sphere", ("Plastic", "DarkPlastic", "Label", "Towel")[k % 4], (1.3, 1.3, 1.3), (rng.uniform(0, 90), rng.uniform(0, 360), 0),
              (5.0 + rr * np.cos(a), 1.9 + 0.85 * (k // 5) + rng.uniform(0, 0.3), 8.0 + rr * np.sin(a)))
    # radiator under the window: a row of vertical tubes
    for k in range(34):
        S.add("cylinder", "StainlessRough", (0.32, 7.5, 0.32), (0, 0, 0), (X0 + 1.3, 5.6, -0.4 + 0.4 * k))
    for y in (2.0, 9.2):
        S.add("cylinder", "StainlessRough", (0.45, 14.0, 0.45), (90, 0, 0), (X0 + 1.3, y, 6.2))
    return S


def main():
    os.makedirs(os.path.join(OUT, "textures"), exist_ok=True)
    if os.path.isdir(REF):
        shutil.copyfile(os.path.join(REF, "bathroom.mtl"), os.path.join(OUT, "bathroom.mtl"))
        shutil.copyfile(os.path.join(REF, "LICENSE.txt"), os.path.join(OUT, "LICENSE.txt"))
        wanted = set()
        for ln in open(os.path.join(REF, "bathroom.mtl"), errors="replace"):
            t = ln.split()
            if t and t[0].startswith("map_") and not ln.lstrip().startswith("#"):
                wanted.add(t[-1].replace("\\", "/"))
        for w in sorted(wanted):
            src = os.path.join(REF, w)
            if os.path.exists(src):
                shutil.copyfile(src, os.path.join(OUT, w))
    for name, fn in PARTS.items():
        p, n, t, f = fn()
        TRI_COUNT[name] = len(f)
        write_ply(os.path.join(OUT, name + ".ply"), p, n, t, f)
    S = build_script()
    head = ["# bathroom2_standin.fa -- generated by tools/gen_bathroom2_standin.py: procedural geometry (models/bathroom2/bathroom.obj is absent from the reference",
            "# checkout) wearing the materials and textures of the reference's own models/bathroom2/bathroom.mtl, seen from the camera of models/bathroom2/bathroom.fa",
            "Camera persp eye -2.520284 15.735250 32.335594 aim -1.976656 14.700628 -2.417851 up 0 1 0 fov 1.768946", "",
            "LoadMaterials bathroom.mtl"]
    open(os.path.join(OUT, "bathroom2_standin.fa"), "w").write("\n".join(head + S.lines) + "\n")
    print("wrote %s: %d instanced objects, %d triangles" % (OUT, S.count, S.tris))


if __name__ == "__main__":
    sys.exit(main())
