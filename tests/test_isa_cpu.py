"""CPU test of invariants of the strip kernels' generated code (no GPU: hipcc cross-compiles gfx950 assembly).

dma_row / dma_warm (cvs_kernels_basis.hip) write M0 inside inline assembly.  M0 is a reserved register for this target: clang ignores --
and warns about -- a clobber of it, so the compiler cannot be TOLD.  What makes this safe is that the compiler itself writes M0 only right
in front of an instruction that reads it, and that these kernels contain no such instruction.  This test keeps that true across
compiler versions and edits: in the assembly of every strip kernel, every instruction that mentions M0 must be one of the two our own
statements emit, and every LDS-DMA load must sit directly behind such a write plus its wait state."""
import hashlib
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "cvsteer_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


def _assembly():
    files = [os.path.join(SRC, n) for n in ("cvs_kernels_basis.hip", "cvs_internal.h", "cvs_device_math.h")]
    key = hashlib.sha1(b"".join(open(f, "rb").read() for f in files)).hexdigest()[:16]
    path = "/tmp/cvs_basis_isa_%s.s" % key
    if not os.path.exists(path):
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-I" + os.path.join(ROOT, "include"),
                        "-I" + SRC, "-S", "--cuda-device-only", files[0], "-o", path + ".tmp"], check=True, stderr=subprocess.DEVNULL)
        os.replace(path + ".tmp", path)
    return open(path).read()


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_m0_is_touched_by_nothing_but_the_lds_dma_statements():
    text = _assembly()
    kernels = re.findall(r"^(_ZN3cvs\w+):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M)
    assert len(kernels) >= 100, len(kernels)
    ours = re.compile(r"^\s*(s_mov_b32 m0, s\d+|s_add_u32 m0, m0, 0x100)\s*(;.*)?$")
    dma = 0
    for name, body in kernels:
        lines = [l for l in body.split("\n") if l.strip() and not l.strip().startswith((";", ".", "//")) and not re.match(r"^\S+:\s*$", l.strip())]
        for i, l in enumerate(lines):
            code = l.split(";")[0]
            if re.search(r"\bm0\b", code):
                assert ours.match(l), (name, l.strip())
            if re.search(r"buffer_load_(dword|ubyte)\b.*\blds\b", code):
                dma += 1
                assert re.match(r"\s*s_nop 0", lines[i - 1]) and ours.match(lines[i - 2]), (name, lines[i - 2:i + 1])
    assert dma > 1000, dma     # every strip kernel stages its rows this way
