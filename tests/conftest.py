"""pytest configuration: `gpu` marks tests that need a real MI355X (run via gpurun)."""
import os
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
PKG = REPO / "cuda-l2_amd"
for p in (str(REPO), str(PKG), str(REPO / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X GPU (run with `-m gpu` on the GPU box)")


def gpu_available() -> bool:
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def repo_dir() -> Path:
    return REPO


@pytest.fixture(scope="session")
def pkg_dir() -> Path:
    return PKG
