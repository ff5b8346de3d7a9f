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
