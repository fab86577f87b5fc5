"""Static hazard check of the shipped gfx950 ISA (tools/isa_hazards.py), CPU only.

hipcc pads compiler-generated consumers of a matrix instruction's destination and does NOT pad inline asm.  Rounds 4 and 5 each shipped (and
found by luck) an inline-asm instruction that read MFMA results too early -- commit 3345069 and the `v_max3_f32` episode of
cnn_fused12rs.h:31-33 --, and k_conv5_wpair reads every accumulator through an inline `v_accvgpr_read_b32` (cnn_conv3p.h:134).  This
test disassembles every code object inside trex_amd/libtrexhip.so and fails on any use of an MFMA destination that sits closer to its MFMA than
the hazard rule allows, on fall-through paths and across every branch edge; a planted fixture proves the checker sees what it is meant to see.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_hazards  # noqa: E402

HIPCC = "/opt/rocm/bin/hipcc"
needs_tools = pytest.mark.skipif(not (os.path.exists(isa_hazards.OBJDUMP) and os.path.exists(HIPCC)), reason="no ROCm LLVM tools here")


@needs_tools
def test_shipped_library_keeps_every_mfma_result_its_wait_states():
    lib = os.path.join(ROOT, "trex_amd", "libtrexhip.so")
    reports, shortest, n_kernels, n_mfma = isa_hazards.check_file(lib)
    # the walk saw the library's matrix code at all: the identity network, the trainer, the fp32 fallbacks
    assert n_kernels >= 30 and n_mfma >= 5000, (n_kernels, n_mfma)
    assert not reports, "MFMA result used too early (inline asm is not padded by hipcc):\n" + "\n".join(reports)
    # the table is the compiler's own: hipcc leaves exactly these distances somewhere in the library, never less
    for mn, need in (("v_mfma_f32_16x16x32_f16", 8), ("v_mfma_f32_32x32x16_f16", 12), ("v_mfma_f32_32x32x2_f32", 18)):
        assert isa_hazards.wait_states(mn) == need
        assert shortest.get(mn, need) >= need, (mn, shortest.get(mn))


@needs_tools
def test_every_inline_asm_valu_mnemonic_of_the_sources_is_known_to_the_walk():
    """the walk checks EVERY instruction, whatever its origin; this pins the list of vector instructions the sources put into asm strings, so that a
    new one is a conscious act (it has to be added here, next to a look at its distance from the MFMAs around it)"""
    import re
    src = os.path.join(ROOT, "trex_amd", "csrc")
    found = set()
    for fn in os.listdir(src):
        if not fn.endswith((".hip", ".h")):
            continue
        text = open(os.path.join(src, fn)).read()
        for m in re.finditer(r'asm\s*(?:volatile)?\s*\(\s*((?:"[^"]*"\s*)+)', text):
            for mn in re.findall(r"\b(v_[a-z0-9_]+)", m.group(1)):
                found.add(mn)
    # cnn_conv3p.h:134 (accumulator reads of conv3), cnn_fused12rs.h:30-32 (pool maxima; on vector results only since the round-5 episode), cnn.hip:795 (the fp16 split)
    assert found <= {"v_accvgpr_read_b32", "v_max3_f32", "v_max_f32", "v_fma_mix_f32"}, found


@needs_tools
@pytest.mark.parametrize("padded", [False, True])
def test_checker_reports_a_planted_hazard_and_accepts_the_padded_form(tmp_path, padded):
    obj = tmp_path / "planted.o"
    cmd = [HIPCC, "-O3", "--offload-arch=gfx950", "--cuda-device-only", "-c", os.path.join(ROOT, "tests", "isa", "planted_hazard.hip"), "-o", str(obj)]
    if padded:
        cmd.insert(1, "-DPADDED")
    subprocess.run(cmd, check=True, capture_output=True)
    reports, shortest, n_kernels, n_mfma = isa_hazards.check_file(str(obj))
    assert n_kernels == 1 and n_mfma == 1
    if padded:
        assert not reports and shortest["v_mfma_f32_32x32x16_f16"] >= 12
    else:
        assert len(reports) == 1 and "v_max3_f32" in reports[0] and "after 0 issue slot" in reports[0]


def test_walk_on_text():
    """the rule itself on hand-written listings: the accumulate chain is free, a reader behind a branch edge is found, s_nop counts its states"""
    def listing(lines):
        return "0000000000001000 <k>:\n" + "".join("\t%s // %012X: 00000000\n" % (ln, 0x1000 + 8 * i) for i, ln in enumerate(lines))
    chain = ["v_mfma_f32_32x32x16_f16 a[0:15], v[0:3], v[4:7], a[0:15]"] * 3
    ok = chain + ["s_nop 7", "s_nop 3", "v_accvgpr_read_b32 v9, a3", "s_endpgm"]
    bad = chain + ["s_nop 7", "s_nop 2", "v_accvgpr_read_b32 v9, a3", "s_endpgm"]
    assert isa_hazards.check_asm(listing(ok))[0] == []
    assert len(isa_hazards.check_asm(listing(bad))[0]) == 1
    # the reader sits at the loop head, the MFMA in front of the backward branch
    loop = ["v_accvgpr_read_b32 v9, a3", "v_add_f32 v9, v9, v9", "v_mfma_f32_32x32x16_f16 a[0:15], v[0:3], v[4:7], a[0:15]",
            "s_cbranch_scc1 65532 <k+0x0>", "s_endpgm"]
    rep = isa_hazards.check_asm(listing(loop))[0]
    assert len(rep) == 1 and "across the branch" in rep[0]
    # nothing falls through an unconditional branch
    jump = ["v_mfma_f32_32x32x16_f16 v[0:15], v[16:19], v[20:23], 0", "s_branch 100 <k+0x200>", "v_max_f32_e32 v1, v2, v3", "s_endpgm"]
    assert isa_hazards.check_asm(listing(jump))[0] == []
