"""CPU: the C-ABI library loads and exports every symbol include/physics_hip.h declares; defaults match
the reference constants; and without a GPU the product path fails loudly (there is no CPU fallback).
No compute entry point is called here."""
import ctypes as C
import os
import re
import subprocess

import pytest

from physics_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "physics_hip.h")


def _declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(phys_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported_and_bound():
    lib = _abi.load_library()
    names = _declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in physics_hip.h but not exported"
        assert n in _abi.PROTOTYPES, f"{n} has no ctypes prototype"
    assert set(_abi.PROTOTYPES) == set(names)


def test_library_carries_gfx950_code_object():
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", _abi.LIB_PATH], capture_output=True, text=True)
    blob = open(_abi.LIB_PATH, "rb").read()
    assert b"gfx950" in blob, "no gfx950 code object embedded"
    assert out.returncode == 0


def test_config_default_matches_reference_constants():
    lib = _abi.load_library()
    cfg = _abi.PhysConfig()
    lib.phys_config_default(C.byref(cfg))
    py = _abi.default_config()
    for name, _ in _abi.PhysConfig._fields_:
        a, b = getattr(cfg, name), getattr(py, name)
        if hasattr(a, "__len__"):
            assert list(a) == list(b), name
        else:
            assert a == b, name
    assert list(cfg.gravity_force) == pytest.approx([0.0, -9.81, 0.0])  # physics.rs:90
    assert list(cfg.gravity_offset) == [0.0, 0.0, 1.5]                  # physics.rs:91
    assert (cfg.cg_max_iterations, cfg.flags) == (1000, 0)              # sle_solver.rs:5
    assert cfg.cg_max_error == pytest.approx(1e-2) and cfg.cg_min_error == pytest.approx(1e-3)
    assert lib.phys_abi_version() == _abi.PHYS_ABI_VERSION


def test_struct_sizes_match_the_c_compiler():
    src = '#include <stdio.h>\n#include "physics_hip.h"\nint main(){printf("%zu %zu %zu %zu", sizeof(phys_config), sizeof(phys_stats), sizeof(phys_profile), sizeof(phys_device_view));}'
    exe = "/tmp/_phys_sizes"
    subprocess.run(["gcc", "-x", "c", "-std=c99", "-I", os.path.join(ROOT, "include"), "-o", exe, "-"], input=src.encode(), check=True)
    sizes = [int(x) for x in subprocess.check_output([exe]).split()]
    assert sizes == [C.sizeof(_abi.PhysConfig), C.sizeof(_abi.PhysStats), C.sizeof(_abi.PhysProfile), C.sizeof(_abi.PhysDeviceView)]


def _gpu_present():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_gpu_present(), reason="needs a machine WITHOUT a GPU")
def test_no_gpu_means_loud_failure_not_a_cpu_fallback():
    import physics_amd
    with pytest.raises(physics_amd.PhysError) as e:
        physics_amd.World()
    assert e.value.code == _abi.PHYS_ERR_NO_DEVICE
    assert "no CPU fallback" in str(e.value) or "HIP" in str(e.value)


def test_product_code_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under physics_amd/ or include/ may reference it."""
    bad = []
    for base in ("physics_amd", "include"):
        for d, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                    txt = open(os.path.join(d, f), errors="ignore").read()
                    if re.search(r"^\s*(from|import)\s+oracle|liboracle|#\s*include\s*[<\"][^>\"]*oracle|\boracle_[a-z_]+\s*\(", txt, flags=re.M):
                        bad.append(os.path.join(d, f))
    assert not bad, bad


def test_independent_gpu_checks_share_no_code_with_the_spec_or_the_oracle():
    """tests/test_gpu_independent.py is the one place where the collision arithmetic is checked against something
    that is NOT compiled from include/spec: it may import numpy, pytest and the product's Python host, nothing else."""
    txt = open(os.path.join(ROOT, "tests", "test_gpu_independent.py")).read()
    mods = set(re.findall(r"^\s*(?:from|import)\s+([A-Za-z0-9_\.]+)", txt, flags=re.M))
    assert mods <= {"numpy", "pytest", "physics_amd"}, mods
    assert "liboracle" not in txt and "include/spec" not in txt.split('"""')[2]


def test_rust_shim_declares_every_header_symbol_once():
    """rust/physics_hip_sys/src/lib.rs is uncompiled source (no cargo / rustc in the image); at least its extern block
    must name exactly the symbols of include/physics_hip.h."""
    txt = open(os.path.join(ROOT, "rust", "physics_hip_sys", "src", "lib.rs")).read()
    rust = re.findall(r"pub fn (phys_[a-z0-9_]+)\s*\(", txt)
    assert sorted(rust) == _declared_symbols(), sorted(set(_declared_symbols()) ^ set(rust))
    assert "pub const PHYS_ABI_VERSION: u32 = %d;" % _abi.PHYS_ABI_VERSION in txt
