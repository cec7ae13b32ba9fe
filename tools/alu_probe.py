"""Platform control for DESIGN.md section 8: a kernel that is a pure function of the thread index (divisions, square
roots, per-lane LDS slices - the shape of the narrow phase) is launched repeatedly, alone and with a bf16 GEMM on a
second stream, and every output is compared bit for bit with the first quiet launch.

    python tools/alu_probe.py [launches] [blocks] [iters] [gemms per launch] [packed]
"""
import ctypes
import os
import subprocess
import sys

import torch

here = os.path.dirname(os.path.abspath(__file__))
lib_path = os.path.join(here, "libalu_probe.so")
if not os.path.exists(lib_path) or os.path.getmtime(lib_path) < os.path.getmtime(os.path.join(here, "alu_probe.hip")):
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "-shared", "--offload-arch=gfx950", "-ffp-contract=off",
                    "-fhip-fp32-correctly-rounded-divide-sqrt", os.path.join(here, "alu_probe.hip"), "-o", lib_path], check=True)
if len(sys.argv) > 1 and sys.argv[1] == "build":
    sys.exit(0)
lib = ctypes.CDLL(lib_path)
lib.alu_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
lib.alu_probe_launch_pk.argtypes = lib.alu_probe_launch.argtypes
launches = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 512
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 400
gemms = int(sys.argv[4]) if len(sys.argv) > 4 else 2
entry = lib.alu_probe_launch_pk if (len(sys.argv) > 5 and sys.argv[5] == "packed") else lib.alu_probe_launch
n = blocks * 128
torch.manual_seed(1)
seed = torch.rand(n, device="cuda") * 4 + 1
mine, side = torch.cuda.Stream(), torch.cuda.Stream()
m = torch.randn(2048, 2048, device="cuda", dtype=torch.bfloat16)


def launch(load):
    out = torch.empty(n, device="cuda")
    if load:
        with torch.cuda.stream(side):
            x = m
            for _ in range(gemms):
                x = x @ m
    rc = entry(out.data_ptr(), seed.data_ptr(), blocks, iters, mine.cuda_stream)
    assert rc == 0, rc
    torch.cuda.synchronize()
    return out


ref = launch(False)
for mode in (False, True):
    bad = 0
    first = None
    for i in range(launches):
        out = launch(mode)
        if not torch.equal(out.view(torch.int32), ref.view(torch.int32)):
            bad += 1
            if first is None:
                idx = torch.nonzero(out.view(torch.int32) != ref.view(torch.int32)).flatten()
                first = (i, int(idx.numel()), idx[:20].tolist(), out[idx[:3]].tolist(), ref[idx[:3]].tolist())
    print(f"{'with GEMM' if mode else 'quiet    '}: {bad} of {launches} launches differ from the reference", first or "")
