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
