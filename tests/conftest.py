import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import binding
    binding.build()
    return binding.load()


def _gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session", autouse=True)
def _torch_gpu_first():
    """On a GPU box, bring PyTorch's HIP context up before libphysics_hip touches the device: tests that open
    torch streams / RCCL groups later then never depend on which other tests ran before them (initialising torch
    after the library had been used failed with hipErrorNoDevice for one particular subset of tests)."""
    if _gpu_available():
        import torch
        torch.cuda.init()
        torch.zeros(1, device="cuda")
    yield


@pytest.fixture(scope="session")
def hip_world_factory():
    """Factory of physics_amd.World on cuda:0. The HIP library must be present: no fallback."""
    import physics_amd

    def make(cfg=None):
        return physics_amd.World(cfg)

    return make
