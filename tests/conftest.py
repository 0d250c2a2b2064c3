"""pytest configuration: markers, path set-up, shared fixtures."""
from __future__ import annotations

import json
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def manifest() -> dict:
    return json.loads((GOLDEN / "MANIFEST.json").read_text())


@pytest.fixture(scope="session", autouse=True)
def _native_built():
    """Build what is missing (oracle .so; product .so) so the suite is self-contained."""
    import oracle_py
    if not oracle_py.ORACLE_SO.exists():
        oracle_py.build()
    from meteor_demod_amd import _capi
    if not _capi.LIB_PATH.exists():
        from meteor_demod_amd.build import build
        build()
    yield


def load_npz(name: str):
    import numpy as np
    return np.load(GOLDEN / f"{name}.npz")


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("a test marked gpu ran without a GPU: the HIP path has no CPU fallback")
    return 0
