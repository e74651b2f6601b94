import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
SCENES = os.path.join(GOLDEN, "scenes")

SHIPPED = [("toy_scene_for_render", "5"), ("toy_scene_for_render_with_materials", "6"), ("Test1", "21"),
           ("Test2", "142"), ("200_rand_1_cam", "1")]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The plain-C FP64 restatement (checker only)."""
    from oracle import bind
    return bind.Oracle()


@pytest.fixture(scope="session")
def ref():
    """The real reference engine; only present in the development container."""
    from oracle import bind
    if not bind.Ref.available():
        pytest.skip("oracle/_ref/libref_rt.so not built (reference sources absent)")
    return bind.Ref()
