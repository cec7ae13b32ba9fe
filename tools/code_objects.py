"""Kernel resource table of the built library: every gfx950 code object inside libphysics_hip.so's .hip_fatbin,
its kernels and their scratch (private segment), LDS, VGPR and SGPR use.

    python tools/code_objects.py [path/to/libphysics_hip.so]

Used by tests/test_build_rules.py for the "no kernel uses scratch memory" rule."""
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _fatbin(lib):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "fat.bin")
        subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={out}", lib, os.path.join(d, "copy.so")], check=True)
        return open(out, "rb").read()


def code_objects(lib):
    """Yield the device ELF images (bytes) of every bundle in the library's fat binary."""
    blob = _fatbin(lib)
    at = blob.find(MAGIC)
    while at >= 0:
        n, = struct.unpack_from("<Q", blob, at + len(MAGIC))
        p = at + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if "amdgcn" in triple and size:
                yield blob[at + off:at + off + size]
        at = blob.find(MAGIC, at + len(MAGIC))


def kernels(lib):
    """[{name, scratch, lds, vgprs, sgprs, spills}] for every kernel in the library."""
    out = []
    for image in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".elf") as f:
            f.write(image)
            f.flush()
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", f.name], check=True, capture_output=True, text=True).stdout
        for block in notes.split("  - .agpr_count:")[1:]:
            def field(key, block=block):
                m = re.search(rf"\.{key}:\s+(\S+)", block)
                return m.group(1) if m else None
            out.append({"name": field("name"), "scratch": int(field("private_segment_fixed_size")),
                        "lds": int(field("group_segment_fixed_size")), "vgprs": int(field("vgpr_count")),
                        "sgprs": int(field("sgpr_count")), "spills": int(field("vgpr_spill_count"))})
    return out


def count_instructions(lib, mnemonics):
    """Occurrences of the given mnemonics in the disassembly of every code object of the library."""
    total = 0
    for image in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".elf") as f:
            f.write(image)
            f.flush()
            text = subprocess.run([f"{LLVM}/llvm-objdump", "-d", f.name], check=True, capture_output=True, text=True).stdout
        total += sum(text.count(m) for m in mnemonics)
    return total


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    args = [a for a in sys.argv[1:] if a != "--check"]
    lib = args[0] if args else os.path.join(here, "..", "physics_amd", "csrc", "libphysics_hip.so")
    if "--check" in sys.argv:
        # the build's own gate (csrc/Makefile): the two rules of DESIGN.md section 8 hold on the code objects just linked,
        # or the build FAILS - a compiler update that drops the -packed-fp32-ops feature must not pass as a warning
        packed = count_instructions(lib, ("v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32"))
        bad = [(k["name"], k["scratch"], k["spills"]) for k in kernels(lib) if k["scratch"] or k["spills"]]
        if packed or bad:
            print(f"code object rules violated: {packed} packed fp32 arithmetic instructions; scratch / spills: {bad}", file=sys.stderr)
            sys.exit(1)
        sys.exit(0)
    ks = kernels(lib)
    for k in sorted(ks, key=lambda k: k["name"]):
        short = subprocess.run(["c++filt", k["name"]], capture_output=True, text=True).stdout.split("(")[0].strip()
        print(f"{short[:60]:60s} scratch {k['scratch']:4d}  lds {k['lds']:7d}  vgprs {k['vgprs']:4d}  sgprs {k['sgprs']:4d}  spills {k['spills']}")
    print(f"{len(ks)} kernels; packed fp32 arithmetic instructions:", count_instructions(lib, ("v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32")))
