"""pytest wiring: marker registration + import paths.

``oracle/`` is importable from tests only (it is the checker, never the product);
``maskcyclegan-vc_amd/`` is the product root (mirrors the reference repo root, so
``import mask_cyclegan_vc`` resolves to the HIP-backed drop-in).
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("oracle", "maskcyclegan-vc_amd"):
    p = os.path.join(ROOT, sub)
    if p not in sys.path:
        sys.path.insert(0, p)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
