#!/usr/bin/env python3
"""ISA lint: every VALU / memory read of an MFMA result must be >= passes+2 wait states after the MFMA on EVERY path.

gfx950 does not interlock these reads; hipcc's hazard recognizer inserts the s_nops, and we have seen one layout (a conditional
branch straight after the last MFMA of a chain, see DESIGN.md "score kernel: a compiler hazard") where the taken path got
9 wait states instead of 18 and the filter read stale accumulators.  This walks the compiled ISA of every kernel in csrc/
and follows both sides of every branch.  Usage: python scripts/lint_mfma_hazard.py [file.hip ...]   (exit 1 on a finding)
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# passes on gfx950.  The 16-bit forms CDNA3 already had run at CDNA3's rate there (1 307 TF dense on 304 CUs x 4 SIMDs x 2.1 GHz = 512 FLOP per
# clock and SIMD: 16x16x16 = 8 192 FLOP = 16 cycles = 4 passes; 32x32x8 = 8 passes) -- half of what gfx90a needed, and what hipcc's own
# hazard recognizer pads for (7 = 4 + 3 states behind v_mfma_f32_16x16x16_bf16 in csrc/enc_tile.hip); CDNA4's double-K forms take the same passes.
PASSES = {"32x32x2_f32": 16, "16x16x4_f32": 8, "32x32x1_2b_f32": 16, "16x16x1_4b_f32": 8, "4x4x1_16b_f32": 2,
          "32x32x8_f16": 8, "16x16x16_f16": 4, "32x32x8_bf16": 8, "16x16x16_bf16": 4, "32x32x16_bf16": 8,
          "16x16x32_bf16": 4, "32x32x16_f16": 8, "16x16x32_f16": 4}
# XDL (16-bit input) MFMAs: result -> vector read needs passes + 3 wait states (8 passes: 11); fp32 "SGEMM" MFMAs passes + 2
XDL = ("_bf16", "_f16")
REG = re.compile(r"\b([va])(?:(\d+)|\[(\d+):(\d+)\])")


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(2) is not None:
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(1), r) for r in range(int(m.group(3)), int(m.group(4)) + 1))
    return out


def parse(path):
    """-> {function: [(label_or_None, mnemonic, operands)]}"""
    funcs, cur, name, pend = {}, None, None, None
    for ln in open(path):
        s = ln.split(";")[0].rstrip()
        if not s.strip():
            continue
        m = re.match(r"^([\w.$]+):\s*$", s)
        if m:
            lab = m.group(1)
            if not lab.startswith(".L"):
                name, cur = lab, []
                funcs[name] = cur
            pend = lab
            if cur is not None:
                cur.append((lab, None, ""))
            continue
        t = s.strip()
        if cur is None or t.startswith(".") or t.startswith(";;#"):
            continue
        parts = t.split(None, 1)
        cur.append((None, parts[0], parts[1] if len(parts) > 1 else ""))
    return funcs


def lint_function(name, ins):
    label_at = {l: i for i, (l, mn, _) in enumerate(ins) if l}
    findings = []
    for i, (_, mn, ops) in enumerate(ins):
        if not mn or not mn.startswith("v_mfma"):
            continue
        key = mn[len("v_mfma_f32_"):] if mn.startswith("v_mfma_f32_") else None
        need = PASSES.get(key, 16) + (3 if key and key.endswith(XDL) else 2)
        dst = regs(ops.split(",")[0])
        seen = set()
        stack = [(i + 1, 0)]
        while stack:
            j, w = stack.pop()
            while j < len(ins) and w < need:
                if (j, w) in seen:
                    break
                seen.add((j, w))
                _, m2, o2 = ins[j]
                if m2 is None:
                    j += 1
                    continue
                if m2 == "s_endpgm":
                    break
                if m2 == "s_nop":
                    w += int(o2.strip(), 0) + 1
                    j += 1
                    continue
                if m2 == "s_branch":
                    j = label_at.get(o2.strip(), len(ins))
                    w += 1
                    continue
                if m2.startswith("s_cbranch"):
                    t = label_at.get(o2.strip())
                    if t is not None:
                        stack.append((t, w + 1))
                    w += 1
                    j += 1
                    continue
                if m2.startswith("v_mfma") or m2.startswith("v_smfmac"):
                    # back-to-back MFMA on the same accumulator has its own (shorter, interlocked-by-rule) distance;
                    # a later MFMA overwriting dst also ends this one's window
                    if regs(o2.split(",")[0]) & dst:
                        break
                elif m2.startswith(("v_", "ds_", "global_", "buffer_", "flat_", "scratch_")):
                    if regs(o2) & dst:
                        findings.append((name, i, mn, j, m2 + " " + o2, w, need))
                        break
                w += 1
                j += 1
    return findings


def main():
    srcs = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "recboard_amd", "csrc", "*.hip")))
    bad = 0
    with tempfile.TemporaryDirectory() as td:
        for src in srcs:
            out = os.path.join(td, os.path.basename(src) + ".s")
            # enc_tile.hip is built WITHOUT packed-fp32 instructions (csrc/Makefile: TILE_FLAGS; profiles/r6_handover_notes.txt): the lint compiles it
            # as the product does and checks that none is left -- with them the tile kernels are not reproducible once two waves share a SIMD
            tile = os.path.basename(src) == "enc_tile.hip"
            extra = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"] if tile else []
            cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only"] + extra + [
                   "-I" + os.path.join(ROOT, "recboard_amd", "csrc"), "-I" + os.path.join(ROOT, "include"), src, "-o", out]
            subprocess.run(cmd, check=True)
            if tile:
                n_pk = sum(1 for ln in open(out) if re.match(r"\s+v_pk_(mul|add|fma)_f32\b", ln))
                print("%-24s %5d packed-fp32 instructions" % (os.path.basename(src), n_pk))
                if n_pk:
                    bad += 1
                    print("HAZARD enc_tile.hip: %d v_pk_*_f32 instructions (the Makefile's TILE_FLAGS must reach this file)" % n_pk)
            n_mfma = 0
            for fn, ins in parse(out).items():
                n_mfma += sum(1 for _, m, _ in ins if m and m.startswith("v_mfma"))
                for f in lint_function(fn, ins):
                    bad += 1
                    print("HAZARD %s: %s @%d read by `%s` @%d after %d/%d wait states" % (f[0], f[2], f[1], f[4], f[3], f[5], f[6]))
            print("%-24s %5d mfma checked" % (os.path.basename(src), n_mfma))
    print("findings:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
