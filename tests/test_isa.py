"""Static checks on the generated gfx950 code (no GPU needed: hipcc cross-compiles).

k_gemm8 orders its LDS-DMA ring with COUNTED `s_waitcnt vmcnt(N)`, N = the number of its own
loads younger than the one awaited (loads retire in order).  A register spill inside the K loop
would put loads the count does not know about into that sequence (a reload between two ring
requests makes vmcnt(N) pass one request early) and would cost a scratch round trip per K-tile,
so the build is checked for scratch traffic of either kind there."""
import os
import re
import shutil
import subprocess

import pytest

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dream2real_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def clip_isa(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa") / "clip.s"
    flags = re.search(r"CXXFLAGS\s*=\s*(.*?)\n\s*-Wall", open(os.path.join(CSRC, "Makefile")).read(), re.S)
    assert flags, "Makefile CXXFLAGS not found"
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-DD2R_MARCH_THREADS=768",
           "-DD2R_GEMM_ABLATE=0", "-I" + os.path.join(CSRC, "..", "..", "include"), "-S", "--cuda-device-only",
           "-o", str(out), os.path.join(CSRC, "clip.hip")]
    subprocess.run(cmd, check=True, capture_output=True, timeout=900)
    return open(out).read()


def kernels(isa, prefix):
    for m in re.finditer(r"^(_Z\d+%s\w*):[^\n]*\n(.*?)s_endpgm" % prefix, isa, re.S | re.M):
        yield m.group(1), m.group(2).split("\n")


def mfma_blocks(lines):
    """The basic blocks that contain MFMA instructions = the K loop (the code layout may put the epilogue between
    them in text order, so "between the first and the last MFMA" is not the loop).  Ends of blocks: labels and
    branches."""
    blocks, cur = [], []
    for l in lines:
        t = l.strip()
        if re.match(r"^\.LBB\w+:", t):
            blocks.append(cur)
            cur = []
        cur.append(l)
        if t.startswith("s_cbranch") or t.startswith("s_branch"):
            blocks.append(cur)
            cur = []
    blocks.append(cur)
    return [b for b in blocks if any("v_mfma" in l for l in b)]


def test_no_spill_stores_inside_the_counted_vmcnt_k_loop(clip_isa):
    seen = 0
    for name, lines in kernels(clip_isa, "k_gemm8"):
        assert sum("v_mfma" in l for l in lines) >= 64, name
        body = [l for b in mfma_blocks(lines) for l in b]
        spills = [l for l in body if "scratch_store" in l or "scratch_load" in l]
        assert not spills, f"{name}: scratch traffic inside the K loop: {spills[:3]}"
        seen += 1
    assert seen >= 4          # one instantiation per epilogue kind


def test_k_loop_waits_are_counted_not_drained(clip_isa):
    """inside the K loop every vmcnt wait is one of the hand-placed ones (inline asm: counted, 10 outstanding);
    a compiler-placed vmcnt wait there would mean a load it knows about (a spill reload, an address it re-fetches)
    inside the counted ring."""
    for name, lines in kernels(clip_isa, "k_gemm8"):
        body = [l for b in mfma_blocks(lines) for l in b]
        # (the fp8 kernel's windows hold one batch of two scale loads each: 12)
        assert sum(("vmcnt(12)" if "k_gemm8f" in name else "vmcnt(10)") in l for l in body) >= 8, name
        in_asm, compiler_waits = False, []
        for l in body:
            if "#ASMSTART" in l:
                in_asm = True
            elif "#ASMEND" in l:
                in_asm = False
            elif "vmcnt" in l and not in_asm:
                compiler_waits.append(l.strip())
        assert not compiler_waits, f"{name}: {compiler_waits[:3]}"


def test_fp8_k_loop_owns_its_fragment_registers(clip_isa):
    """k_gemm8f keeps its MFMA operand fragments in v160-v255 by name (inline asm): between the loop's asm statements the compiler
    must not touch those registers, must not spill (scratch loads would join the hand-counted vmcnt queue) and must not copy a
    scale register between its load and the MFMAs that read it (a copy made before the load has landed copies stale data)."""
    seen = 0
    for name, lines in kernels(clip_isa, "k_gemm8f"):
        body = [l for b in mfma_blocks(lines) for l in b]
        assert sum("v_mfma_scale_f32_32x32x64_f8f6f4" in l for l in body) >= 32, name
        in_asm, scale_regs, compiler = False, set(), []
        for l in body:
            t = l.strip()
            if "#ASMSTART" in t:
                in_asm = True
            elif "#ASMEND" in t:
                in_asm = False
            elif in_asm:
                m = re.match(r"global_load_dword (v\d+),", t)
                if m:
                    scale_regs.add(m.group(1))
            elif t and not t.startswith(";") and not t.startswith("."):
                compiler.append(t)
        assert len(scale_regs) >= 4, name
        for t in compiler:
            regs = [int(x) for x in re.findall(r"\bv(\d+)\b", t)]
            for a, b in re.findall(r"v\[(\d+):(\d+)\]", t):
                regs += list(range(int(a), int(b) + 1))
            assert not any(r >= 160 for r in regs), f"{name}: compiler instruction on a fragment register: {t}"
            assert "scratch_" not in t and "v_accvgpr" not in t, f"{name}: {t}"
            assert not any(re.search(r"\b%s\b" % r, t) for r in scale_regs), f"{name}: compiler instruction on a scale register: {t}"
        seen += 1
    assert seen == 3


def test_attention_q_registers_are_not_touched_before_their_wait(clip_isa):
    """k_attention_s loads its Q fragments from inline asm (so that hipcc does not drain the request ring for them inside the
    key-tile loop) and waits for them by hand.  hipcc believes the registers are ready the moment the asm statement ends: any
    instruction it places between the loads and the first vmcnt wait that touches them reads stale data (round 6 saw exactly
    that — v_mov_b64 copies hoisted above the wait — when a branch sat between the two)."""
    seen = 0
    for name, lines in kernels(clip_isa, "k_attention_s"):
        L = [l.strip() for l in lines]
        q = [i for i, l in enumerate(L) if l.startswith("global_load_dwordx4") and "lds" not in l][:4]
        assert len(q) == 4, name
        regs = set()
        for i in q:
            m = re.search(r"global_load_dwordx4 v\[(\d+):(\d+)\]", L[i])
            regs.update(range(int(m.group(1)), int(m.group(2)) + 1))
        assert len(regs) == 16, name
        wait = next(i for i in range(q[-1] + 1, len(L)) if L[i].startswith("s_waitcnt") and "vmcnt" in L[i])
        for i in range(q[-1] + 1, wait):
            l = L[i]
            if not l or l.startswith((";", ".")):
                continue
            touched = any(int(a) <= r <= int(b) for a, b in re.findall(r"v\[(\d+):(\d+)\]", l) for r in regs) or \
                any(int(a) in regs for a in re.findall(r"\bv(\d+)\b", l))
            assert not touched, f"{name}: `{l}` touches a Q register before the first vmcnt wait"
        seen += 1
    assert seen >= 4
