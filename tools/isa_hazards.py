"""Static check of the SHIPPED gfx950 ISA for one hazard class: a matrix-core (MFMA) result read -- or overwritten -- too early.

Why: hipcc's hazard recognizer pads compiler-generated consumers of an MFMA's destination, but NOT instructions that come out of an
inline-asm string (cdna_hip_programming.md 5.7 item 2).  Rounds 4 and 5 each found such a site by luck (`v_max3_f32` on conv1's
accumulators, commit 3345069; `k_conv5_wpair` reads every accumulator through an inline `v_accvgpr_read_b32`).  This tool disassembles
every gfx950 code object inside libtrexhip.so (or a given object / code-object file) and walks each function:

  for every instruction that is not the MFMA's own accumulate chain and that touches (reads OR writes) a register an earlier MFMA wrote,
  the number of issue slots between the two (every instruction = 1 wait state, `s_nop N` = N + 1) must be at least PASSES-dependent:
        v_mfma_f32_16x16x32_f16 8, v_mfma_f32_32x32x16_f16 / _bf16 12, v_mfma_f32_32x32x2_f32 18, ...  (the table `PASSES` below)
  An MFMA that takes the whole destination as its C operand (the accumulate chain) needs none.

The walk is linear in address order (fall-through paths) plus every backward branch's wrap-around (loop end -> loop head).  Distances are
a LOWER bound of what the hardware sees (a taken branch, a barrier or a wait only add time), so a report is a real too-short distance in
the instruction stream.  Compiler-generated code satisfies the rule by construction; a report therefore points at an inline-asm site (or
at a table entry below that is too strict -- the tool prints the shortest distance it saw per MFMA mnemonic so that can be judged).

   python tools/isa_hazards.py [file ...]        default: trex_amd/libtrexhip.so; exit status 1 when a site is reported
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"

# (mnemonic, passes of 4 clocks, XDL?) of the matrix instructions on gfx950; anything else counts as a 16-pass XDL instruction (the most
# conservative).  Wait states an MFMA's destination needs in front of any other use (LLVM's GCNHazardRecognizer for gfx950; they are also
# exactly the shortest distances hipcc itself leaves in this library: 8 behind v_mfma_f32_16x16x32_f16, 12 behind v_mfma_f32_32x32x16_f16 /
# _bf16, 18 behind v_mfma_f32_32x32x2_f32 -- the tool prints them, tests/test_isa_hazards.py asserts them):
#     XDL (f16 / bf16 / i8 inputs): passes + 3, + 1 on gfx950 unless 2-pass        fp32-input MFMA: passes + 2
PASSES = [
    (re.compile(r"^v_mfma_f32_32x32x16_(f16|bf16)"), 8, True),
    (re.compile(r"^v_mfma_f32_16x16x32_(f16|bf16)"), 4, True),
    (re.compile(r"^v_mfma_f32_32x32x8_?(f16|bf16)"), 8, True),
    (re.compile(r"^v_mfma_f32_16x16x16_?(f16|bf16)"), 4, True),
    (re.compile(r"^v_mfma_f32_32x32x2_?f32"), 16, False),
    (re.compile(r"^v_mfma_f32_16x16x4_?f32"), 8, False),
    (re.compile(r"^v_mfma_f32_4x4x1_?(16b_)?f32"), 2, False),
    (re.compile(r"^v_mfma_f32_4x4x4_?(16b_)?f16"), 2, True),
]


def wait_states(mn):
    for rx, p, xdl in PASSES:
        if rx.match(mn):
            return (p + 3 + (1 if p != 2 else 0)) if xdl else p + 2
    return 16 + 4


REG = re.compile(r"\b([av])(?:\[(\d+):(\d+)\]|(\d+)\b)")
INSN = re.compile(r"^\s+([a-z_0-9]+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")
FUNC = re.compile(r"^[0-9a-f]+ <(.+)>:")
BRANCH_T = re.compile(r"<[^>+]+\+0x([0-9a-fA-F]+)>")


def code_objects(path):
    """gfx950 code objects inside `path`: a host .so / .o with clang offload bundles, or a bare AMDGPU ELF."""
    data = open(path, "rb").read()
    out = []
    pos = 0
    while True:
        i = data.find(MAGIC, pos)
        if i < 0:
            break
        n = struct.unpack_from("<Q", data, i + 24)[0]
        off = i + 32
        for _ in range(n):
            o, sz, ts = struct.unpack_from("<QQQ", data, off)
            off += 24
            triple = data[off:off + ts].decode(errors="replace")
            off += ts
            if "gfx950" in triple and sz:
                out.append(data[i + o:i + o + sz])
        pos = i + 24
    if not out and data[:4] == b"\x7fELF":
        out.append(data)
    return out


def disassemble(blob):
    with tempfile.NamedTemporaryFile(suffix=".elf") as f:
        f.write(blob)
        f.flush()
        return subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", f.name], capture_output=True, text=True, check=True).stdout


def regs(text):
    s = set()
    for m in REG.finditer(text):
        k = m.group(1)
        if m.group(4) is not None:
            s.add((k, int(m.group(4))))
        else:
            for r in range(int(m.group(2)), int(m.group(3)) + 1):
                s.add((k, r))
    return s


def parse_functions(asm):
    funcs, cur = [], None
    for line in asm.splitlines():
        f = FUNC.match(line)
        if f:
            cur = (f.group(1), [])
            funcs.append(cur)
            continue
        m = INSN.match(line)
        if m and cur is not None:
            mn, ops, addr = m.group(1), m.group(2), int(m.group(3), 16)
            cur[1].append((addr, mn, ops))
    return funcs


def states(mn, ops):
    if mn == "s_nop":
        try:
            return int(ops.strip(), 0) + 1
        except ValueError:
            return 1
    return 1


def check_stream(fname, insns, reports, shortest, tag=""):
    """insns: list of (addr, mnemonic, operands) in issue order."""
    live = []          # (dest registers, required states, states since, mnemonic, addr, dest operand text)
    for addr, mn, ops in insns:
        touched = regs(ops)
        is_mfma = mn.startswith("v_mfma") or mn.startswith("v_smfmac")
        first = ops.split(",")[0].strip() if ops else ""
        if touched:
            for dest, need, since, pmn, paddr, ptext in live:
                if not (touched & dest):
                    continue
                if is_mfma:
                    parts = [p.strip() for p in ops.split(",")]
                    cregs = regs(parts[3]) if len(parts) > 3 else set()
                    abregs = (regs(parts[1]) if len(parts) > 1 else set()) | (regs(parts[2]) if len(parts) > 2 else set())
                    dregs = regs(parts[0])
                    if not (abregs & dest) and (cregs == dest or not (cregs & dest)) and (dregs == dest or not (dregs & dest)):
                        continue                      # the whole destination taken as C (accumulate chain, or summed into another tile), or overwritten whole
                key = pmn.split(" ")[0]
                shortest[key] = min(shortest.get(key, 1 << 30), since)
                if since < need:
                    reports.append("%s%s: %s %s at %#x touches %s written by %s at %#x after %d issue slot(s), %d needed"
                                   % (fname[:90], tag, mn, ops, addr, ptext, pmn, paddr, since, need))
        if mn in ("s_branch", "s_endpgm", "s_setpc_b64", "s_swappc_b64", "s_trap"):
            live = []                                 # nothing falls through
            continue
        # age, retire, then add this instruction's own destination
        st = states(mn, ops)
        live = [(d, n, s + st, pm, pa, pt) for (d, n, s, pm, pa, pt) in live if s + st < 24]
        if is_mfma:
            d = regs(first)
            live = [x for x in live if not (x[0] & d)]           # the newer write is the one a later reader waits for
            live.append((d, wait_states(mn), 0, mn, addr, first))
        elif touched:
            # a later non-matrix write to the same registers ends the matrix instruction's claim on them (it was checked above)
            wd = regs(first) if not mn.startswith(("s_", "ds_write", "global_store", "buffer_store", "scratch_store", "flat_store")) else set()
            if wd:
                live = [(d, n, s, pm, pa, pt) for (d, n, s, pm, pa, pt) in live if not (d & wd)]


def check_asm(asm):
    reports, shortest = [], {}
    nfun = nmfma = 0
    for fname, insns in parse_functions(asm):
        if not any(mn.startswith("v_mfma") for _, mn, _ in insns):
            continue
        nfun += 1
        nmfma += sum(1 for _, mn, _ in insns if mn.startswith("v_mfma"))
        check_stream(fname, insns, reports, shortest)
        # every branch edge: the instructions in front of the branch, then the instructions from its target on (loops and forward jumps)
        index = {a: i for i, (a, _, _) in enumerate(insns)}
        base = insns[0][0]
        for i, (addr, mn, ops) in enumerate(insns):
            if not mn.startswith("s_cbranch") and mn != "s_branch":
                continue
            t = BRANCH_T.search(ops)
            if not t:
                continue
            tgt = base + int(t.group(1), 16)
            j = index.get(tgt)
            if j is None:
                continue
            lo = max(0, i - 24)
            for q in range(i - 1, lo - 1, -1):          # not past an instruction nothing falls through
                if insns[q][1] in ("s_branch", "s_endpgm", "s_setpc_b64"):
                    lo = q + 1
                    break
            head = insns[lo:i]                          # (the branch itself is one issue slot)
            wrap = head + [(addr, "s_nop", "0")] + insns[j:j + 24]
            sub, sh2 = [], {}
            check_stream(fname, wrap, sub, sh2, " (across the branch at %#x)" % addr)
            # only pairs that straddle the edge are new: the consumer lies behind the target, the producer in front of the branch
            tail_addrs = set(a for a, _, _ in insns[j:j + 24])
            head_addrs = set(a for a, _, _ in head)
            for r in sub:
                m = re.search(r" at (0x[0-9a-f]+) touches .* at (0x[0-9a-f]+) after", r)
                if m and int(m.group(1), 16) in tail_addrs and int(m.group(2), 16) in head_addrs:
                    reports.append(r)
    return reports, shortest, nfun, nmfma


def check_file(path):
    allr, short, nf, nm = [], {}, 0, 0
    for blob in code_objects(path):
        r, s, f, m = check_asm(disassemble(blob))
        allr += r
        nf += f
        nm += m
        for k, v in s.items():
            short[k] = min(short.get(k, 1 << 30), v)
    return sorted(set(allr)), short, nf, nm


def main():
    paths = sys.argv[1:] or [os.path.join(ROOT, "trex_amd", "libtrexhip.so")]
    bad = 0
    for p in paths:
        rep, short, nf, nm = check_file(p)
        print("%s: %d kernels with matrix instructions, %d MFMAs, %d hazard site(s)" % (p, nf, nm, len(rep)))
        for k in sorted(short):
            print("   shortest distance MFMA result -> other use: %-32s %2d issue slots (needed %d)" % (k, short[k], wait_states(k)))
        for r in rep:
            print("   HAZARD " + r)
        bad += len(rep)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
