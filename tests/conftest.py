import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def table():
    from fermat_amd import scene
    t = np.fromfile(os.path.join(scene.DATA_DIR, "glossy_reflectance.dat"), np.float32)
    assert t.size == 32 ** 4
    return t


@pytest.fixture(scope="session")
def olib():
    from oracle import binding
    return binding.lib()


@pytest.fixture(scope="session")
def cornell():
    from fermat_amd import scene
    return scene.cornell_box("CornellBox-JP")


@pytest.fixture(scope="session")
def cornell_glossy():
    from fermat_amd import scene
    return scene.cornell_box("CornellBox-Glossy")


@pytest.fixture(scope="session")
def standin_small():
    from fermat_amd import scene
    return scene.bathroom_standin(0.08)


def make_glow_panel_scene(tmp_dir, texture, ke=(4.0, 3.0, 2.0), scaling=(3.0, 1.0), with_map=True):
    """CornellBox-JP plus a tessellated emissive panel whose emission is modulated by `texture` (H,W,3 uint8, written as TGA)
    through map_Ke -s sx sy.  Returns the pre-processed Scene (loaded through the .fa front-end)."""
    import os
    import numpy as np
    from fermat_amd import scene
    tmp_dir = str(tmp_dir)
    h, w = texture.shape[:2]
    hdr = bytearray(18); hdr[2] = 2; hdr[12] = w & 255; hdr[13] = w >> 8; hdr[14] = h & 255; hdr[15] = h >> 8; hdr[16] = 24
    open(os.path.join(tmp_dir, "glow.tga"), "wb").write(bytes(hdr) + np.ascontiguousarray(texture[..., ::-1]).tobytes())
    with open(os.path.join(tmp_dir, "panel.mtl"), "w") as f:
        f.write("newmtl glow\nKd 0.2 0.2 0.2\nKe %g %g %g\n" % ke)
        if with_map:
            f.write("map_Ke -s %g %g glow.tga\n" % scaling)
    n = 4
    with open(os.path.join(tmp_dir, "panel.obj"), "w") as f:
        f.write("mtllib panel.mtl\n")
        for j in range(n + 1):
            for i in range(n + 1):
                f.write("v %g %g %g\n" % (-0.5 + i / n, 0.4 + 0.8 * j / n, -0.99))
                f.write("vt %g %g\n" % (i / n * 0.9 + 0.05, j / n * 1.7 - 0.3))
        f.write("vn 0 0 1\ng panel\nusemtl glow\n")
        for j in range(n):
            for i in range(n):
                a = j * (n + 1) + i + 1; b = a + 1; c = a + n + 2; d = a + n + 1
                f.write("f %d/%d/1 %d/%d/1 %d/%d/1 %d/%d/1\n" % (a, a, b, b, c, c, d, d))
    cornell = os.path.join(scene.DATA_DIR, "scenes", "CornellBox")
    with open(os.path.join(tmp_dir, "glow.fa"), "w") as f:
        f.write("LoadScene %s/CornellBox-JP.obj\nLoadScene panel.obj\n" % cornell)
    s = scene.load_scene(os.path.join(tmp_dir, "glow.fa"))
    s.camera = scene.load_camera(os.path.join(cornell, "camera-frontal.txt"))
    return s


def grazing_rays(scn, n, seed, ray_dtype, shadow=False):
    """Rays that lie (almost) IN the plane of a triangle of the scene and pass through it: det -> 0 in Moller-Trumbore, the computed t is the quotient of two
    cancellations.  These are the rays the intersector's box clause exists for: without it, whether such a triangle is tested -- and so the answer
    -- depends on the tree."""
    rng = np.random.default_rng(seed)
    vi = scn.vertex_indices[:, :3]; P = scn.vertex_data[:, :3].astype(np.float64)
    k = rng.integers(0, len(vi), n)
    v0, v1, v2 = P[vi[k, 0]], P[vi[k, 1]], P[vi[k, 2]]
    e1, e2 = v1 - v0, v2 - v0
    nrm = np.cross(e1, e2); ln = np.linalg.norm(nrm, axis=1, keepdims=True); nrm = nrm / np.maximum(ln, 1e-300)
    b = rng.random((n, 2)); flip = b.sum(1) > 1; b[flip] = 1 - b[flip]
    c = v0 + b[:, :1] * e1 + b[:, 1:] * e2                                  # a point inside the triangle
    a = rng.random((n, 1)) * 2 * np.pi
    e1n = e1 / np.maximum(np.linalg.norm(e1, axis=1, keepdims=True), 1e-300)
    w = np.cos(a) * e1n + np.sin(a) * np.cross(nrm, e1n)                    # an in-plane direction
    lo, hi = scn.bbox; ext = float(np.max(np.asarray(hi, np.float64) - np.asarray(lo, np.float64)))
    dist = ext * (0.05 + 1.5 * rng.random((n, 1)))
    off = ext * rng.choice([0.0, 1e-8, -1e-8, 1e-7, -1e-7, 1e-6, -1e-6, 1e-5, -1e-5, 1e-4], (n, 1))
    o = c - dist * w + off * nrm
    rays = np.zeros(n, ray_dtype)
    rays["origin"] = o.astype(np.float32)
    if shadow:
        rays["dir"] = ((c + dist * w * rng.choice([1e-4, 0.3, 1.0], (n, 1))) - o).astype(np.float32)          # unnormalised, ending just behind / well behind the point
        rays["tmax"] = 0.9999
        rays["mask"] = np.where(np.arange(n) % 2 == 0, 0x2, 0x1).astype(np.uint32)
    else:
        d = (c - o); ln = np.linalg.norm(d, axis=1, keepdims=True)
        d = np.where(ln > 0, d / np.maximum(ln, 1e-300), np.float64([0.0, 0.0, 1.0]))          # (zero-area triangles have no plane to aim along)
        rays["dir"] = d.astype(np.float32)
        rays["mask"] = np.float32(1e-3).view(np.uint32)
        rays["tmax"] = 1e8
    return rays
