"""Build rules of the HIP library that the parity claims lean on, checked on the built code objects
themselves - no GPU needed.

Rule: no kernel uses scratch (private segment) memory. The narrow phase once kept two shapes and a manifold in 256
bytes of scratch; with a bf16 GEMM of another library running on a second stream, whole 16-lane groups of that
scratch came back wrong (tools/race_probe.py: manifolds missing in multiples of 16). Everything a lane indexes at
run time now lives in LDS, and this test keeps it that way."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
LIB = os.path.join(ROOT, "physics_amd", "csrc", "libphysics_hip.so")


@pytest.fixture(scope="module")
def kernel_table():
    import code_objects
    if not os.path.exists(LIB):
        import __graft_entry__
        __graft_entry__.build()
    table = code_objects.kernels(LIB)
    assert len(table) >= 40, "the fat binary should hold every kernel of the library"
    return table


def test_no_kernel_uses_scratch_memory(kernel_table):
    offenders = [(k["name"], k["scratch"], k["spills"]) for k in kernel_table if k["scratch"] or k["spills"]]
    assert not offenders, f"kernels with scratch memory or register spills: {offenders}"


def test_no_packed_fp32_arithmetic():
    """Rule: no v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 anywhere (csrc/Makefile: -packed-fp32-ops). Next to the MFMA
    waves of another stream's bf16 GEMM these returned, for 16 lanes at a time, results off in the 5th digit - the
    narrow phase's clip planes moved by 1e-4 (tools/frozen_probe.py)."""
    import code_objects
    if not os.path.exists(LIB):
        import __graft_entry__
        __graft_entry__.build()
    found = code_objects.count_instructions(LIB, ("v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32"))
    assert found == 0, f"{found} packed fp32 arithmetic instructions in libphysics_hip.so"


def test_lds_fits_one_workgroup(kernel_table):
    # 160 KB of LDS per CU on gfx950; a static allocation above 64 KB needs the whole-CU budget and must stay under it
    for k in kernel_table:
        assert k["lds"] <= 160 * 1024, k
