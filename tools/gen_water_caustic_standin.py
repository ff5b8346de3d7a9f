#!/usr/bin/env python3
"""Writes fermat_amd/data/scenes/water_caustic_standin/: the stand-in for BASELINE configs[4]'s scene, models/water_caustic (VERDICT r4 task 1).

What the reference checkout HAS of water_caustic is used as it is: water_caustic.mtl (Green / Red / Silver / Water / White / Light / Light2 -- the Water material is
`Kd 0, Ns 1024, d 0, Ks 1`: a nearly specular, fully transmissive GGX lobe; the two emitters radiate 8000 6800 4400 and 6000 7200 8000), camera.txt, the camera
line of water_caustic.fa and readme.txt (CC0, B. Bitterli) -- copied as data.  What it LACKS is water_caustic.obj (.MISSING_LARGE_BLOBS): the geometry here is
procedural -- a closed Cornell-style room (Green left, Red right, White elsewhere) seen from water_caustic.fa's own camera, a pool behind a low wall whose
surface is a wavy height field in `Water` (a periodic 128 x 128 tile instanced 5 x 5 times: 819 200 triangles), Silver spheres and blocks in and around the
pool, pool steps, and two SMALL quads in `Light` / `Light2` above the water -- so that the paths the bidirectional tracer exists for are there: light -> water
(reflected or transmitted, roughness 1/1024) -> wall / pool floor -> eye, which next-event estimation cannot sample through the water surface.
NB the reference's loader leaves the Water material's index of refraction at its default 1 (the .mtl has no Ni line; MeshBase.cpp:354-412), so in Fermat this
scene's water transmits straight through and its caustics are the reflected ones; that is the reference's behaviour and is kept.

  python tools/gen_water_caustic_standin.py          # deterministic: rewrites the committed files bit for bit
"""
import os
import shutil
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_bathroom2_standin as parts      # noqa: E402  (the part meshes and the .fa instancing helper are shared)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "fermat_amd", "data", "scenes", "water_caustic_standin")
REF = "/root/reference/models/water_caustic"

# the room: the camera of water_caustic.fa stands at (2.8, 3.49, 11.13) and looks down -z, slightly downwards
X0, X1, Y1, Z0, Z1 = -3.7, 9.3, 7.5, -7.0, 12.5
POOL_Z1 = 7.5            # the pool reaches from the back wall to a low wall at z = 7.5 .. 7.9, just in front of the camera: the water fills the lower half of the view
WATER_Y = 1.0
TILES_X, TILES_Z = 5, 5


def part_water(n=128):
    """one periodic tile of the water surface: unit square in xz, centred, y = a sum of sines with integer wave numbers (so that instances tile seamlessly),
    analytic normals"""
    u, v = np.meshgrid(np.linspace(0.0, 1.0, n + 1), np.linspace(0.0, 1.0, n + 1), indexing="ij")
    waves = ((1, 0, 0.030, 0.3), (0, 1, 0.024, 1.1), (1, 1, 0.018, 2.0), (2, -1, 0.014, 0.7), (3, 2, 0.009, 4.2), (-2, 3, 0.008, 5.1), (5, 1, 0.005, 3.3), (4, -4, 0.004, 1.9),
             (7, 3, 0.0025, 0.2), (-6, 7, 0.002, 2.6))
    h = np.zeros_like(u); hu = np.zeros_like(u); hv = np.zeros_like(u)
    for m, k, a, ph in waves:
        arg = 2.0 * np.pi * (m * u + k * v) + ph
        h += a * np.sin(arg); hu += a * np.cos(arg) * 2.0 * np.pi * m; hv += a * np.cos(arg) * 2.0 * np.pi * k
    P = np.stack([u - 0.5, h, -(v - 0.5)], 2)
    p, _, tri = parts.grid_mesh(P, n, n)
    # P = (u - 0.5, h(u, v), 0.5 - v): dP/du = (1, hu, 0), dP/dv = (0, hv, -1); the upward normal is dP/dv x dP/du = (-hu, 1, hv) up to scale
    nr = np.stack([-hu, np.ones_like(hu), hv], 2).reshape(-1, 3)
    nr /= np.linalg.norm(nr, axis=1, keepdims=True)
    return p, nr, parts.lattice_uv(n, n, 1.0, 1.0), tri


PARTS = {"quad": parts.part_quad, "box": parts.part_box, "sphere": parts.part_sphere, "cylinder": parts.part_cylinder, "torus": parts.part_torus, "water": part_water}


def build_script():
    S = parts.Script()
    cx, cz = 0.5 * (X0 + X1), 0.5 * (Z0 + Z1)
    W, D = X1 - X0, Z1 - Z0
    # the shell (all quads face inwards)
    S.add("quad", "White", (W, 1, D), (0, 0, 0), (cx, 0.0, cz))                        # floor
    S.add("quad", "White", (W, 1, D), (180, 0, 0), (cx, Y1, cz))                       # ceiling
    S.add("quad", "White", (W, 1, Y1), (90, 0, 0), (cx, Y1 / 2, Z0))                   # back wall
    S.add("quad", "White", (W, 1, Y1), (-90, 0, 0), (cx, Y1 / 2, Z1))                  # wall behind the camera
    S.add("quad", "Green", (Y1, 1, D), (0, 0, -90), (X0, Y1 / 2, cz))                  # left wall
    S.add("quad", "Red", (Y1, 1, D), (0, 0, 90), (X1, Y1 / 2, cz))                     # right wall
    # the pool: a low wall in front, the water surface as TILES_X x TILES_Z instances of the periodic tile
    S.add("box", "White", (W, 1.3, 0.4), (0, 0, 0), (cx, 0.65, POOL_Z1 + 0.2))
    tw, td = W / TILES_X, (POOL_Z1 - Z0) / TILES_Z
    for i in range(TILES_X):
        for j in range(TILES_Z):
            S.add("water", "Water", (tw, 1.0, td), (0, 0, 0), (X0 + (i + 0.5) * tw, WATER_Y, Z0 + (j + 0.5) * td))
    # steps into the pool on the right, under water
    for k in range(4):
        S.add("box", "White", (2.2, 0.25, 1.1), (0, 0, 0), (X1 - 1.1, 0.125 + 0.25 * k, POOL_Z1 - 0.55 - 1.1 * k))
    # Silver: three spheres in the water (one floating half-way out, two resting on the pool floor), a torus and two blocks at the pool's edge
    S.add("sphere", "Silver", (2.0, 2.0, 2.0), (0, 0, 0), (0.4, 1.0, -3.2))
    S.add("sphere", "Silver", (1.3, 1.3, 1.3), (0, 0, 0), (4.3, 0.65, -1.0))
    S.add("sphere", "Silver", (0.9, 0.9, 0.9), (0, 0, 0), (6.6, 0.45, -4.6))
    S.add("torus", "Silver", (1.8, 1.8, 1.8), (0, 0, 0), (2.4, 0.98, 2.6))
    S.add("box", "Silver", (1.2, 2.4, 1.2), (0, 25, 0), (-1.9, 1.2, 4.2))
    S.add("box", "Silver", (1.6, 0.9, 1.0), (0, -15, 0), (6.4, 0.45, 4.8))
    S.add("cylinder", "Silver", (0.5, 3.0, 0.5), (0, 0, 0), (8.2, 1.5, -6.0))
    # the two emitters: small quads facing down, well above the water (water_caustic.mtl: Light 8000 6800 4400, Light2 6000 7200 8000)
    S.add("quad", "Light", (0.16, 1, 0.16), (180, 0, 0), (0.9, 6.3, -2.4))
    S.add("quad", "Light2", (0.12, 1, 0.12), (180, 0, 0), (5.6, 5.6, -4.4))
    # their housings (so that the emitters do not float): white blocks above them
    S.add("box", "White", (0.5, 0.25, 0.5), (0, 0, 0), (0.9, 6.45, -2.4))
    S.add("box", "White", (0.4, 0.25, 0.4), (0, 0, 0), (5.6, 5.75, -4.4))
    S.add("cylinder", "White", (0.06, Y1 - 6.55, 0.06), (0, 0, 0), (0.9, 0.5 * (Y1 + 6.55), -2.4))
    S.add("cylinder", "White", (0.06, Y1 - 5.85, 0.06), (0, 0, 0), (5.6, 0.5 * (Y1 + 5.85), -4.4))
    return S


def main():
    os.makedirs(OUT, exist_ok=True)
    if os.path.isdir(REF):
        for name in ("water_caustic.mtl", "camera.txt", "readme.txt"):
            shutil.copyfile(os.path.join(REF, name), os.path.join(OUT, name))
    for name, fn in PARTS.items():
        p, n, t, f = fn()
        parts.TRI_COUNT[name] = len(f)
        parts.write_ply(os.path.join(OUT, name + ".ply"), p, n, t, f)
    S = build_script()
    head = ["# water_caustic_standin.fa -- generated by tools/gen_water_caustic_standin.py: procedural geometry (models/water_caustic/water_caustic.obj is absent from the",
            "# reference checkout) wearing the materials of the reference's own models/water_caustic/water_caustic.mtl, seen from the camera of models/water_caustic/water_caustic.fa",
            "Camera persp eye 2.800000 3.487043 11.134270 aim 2.800000 2.043976 -3.464071 up 0 1 0 fov 0.873", "",
            "LoadMaterials water_caustic.mtl"]
    open(os.path.join(OUT, "water_caustic_standin.fa"), "w").write("\n".join(head + S.lines) + "\n")
    print("wrote %s: %d instanced objects, %d triangles" % (OUT, S.count, S.tris))


if __name__ == "__main__":
    sys.exit(main())
