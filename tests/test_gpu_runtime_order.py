"""GPU: the library first, PyTorch second, in a FRESH process - the order that failed in round 1 with
hipErrorNoDevice ("no ROCm-capable device is detected") at torch.cuda.Stream(). Cause: PyTorch ships its own ROCm
runtime (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7) next to the system's; the library loaded first pulled in
/opt/rocm's, torch then brought a second runtime into the process and that one could not initialise. The loader
(physics_amd._abi.share_rocm_runtime_with_torch) now maps ONE runtime for both, whichever comes first."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_loader_maps_one_rocm_runtime_in_either_order():
    """CPU: no GPU needed to see which runtime files a process maps."""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import physics_amd._abi as a\n"
            "a.load_library(); first = a.rocm_runtime_mapped()\n"
            "import torch; both = a.rocm_runtime_mapped()\n"
            "print(len(first), len(both), first == both)\n" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.stdout.split() == ["1", "1", "True"], out.stdout + out.stderr


@pytest.mark.gpu
def test_library_first_then_torch_in_a_fresh_process():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "nodevice_probe.py")], cwd=ROOT, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0 and "after torch:" in out.stdout and "sum 8.0" in out.stdout, out.stdout + out.stderr
