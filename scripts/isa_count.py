#!/usr/bin/env python3
"""Instruction mix of a kernel's hot loop, read from the code object that ships in libvo_hip.so.

usage: isa_count.py [object file] [kernel name substring] [anchor instruction] > profiles/r2_score_fund_isa.json

Unbundles the gfx950 code object of visual_odometry_amd/csrc/_obj/vo_score.o (what the Makefile links
into the library), disassembles it with llvm-objdump, finds inside `score_kernel<0>` (the F count kernel)
the innermost backward-branch loop that holds `v_alignbit_b32` but no `v_min_f32` and the most `v_fma`/`v_fmac`
instructions (the unrolled tier-1a loop: the one-sided sure-outlier test every evaluation goes through; the loop
with v_min_f32 is the complete sign-bit filter kept for ragged tails) and counts its instructions by issue class.  One `ds_read_b128` = one correspondence, so
"per evaluation" = per ds_read_b128 of the loop body.  bench.py multiplies the VALU figure by the
evaluations of a launch to price the kernel against the VALU issue rate (roofline.bound = "valu-issue").
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def disassemble(obj):
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "k.co")
        subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                               f"--input={fat}", f"--output={co}", "--unbundle"])
        return subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", co], text=True)


def kernel_body(asm, needle):
    out, on = [], False
    for line in asm.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:", line)
        if m:
            on = needle in m.group(1)
            continue
        if on and "//" in line:
            text, tail = line.split("//", 1)
            addr = int(tail.split(":")[0].strip(), 16)
            parts = text.split()
            if parts:
                out.append((addr, parts[0], text.strip()))
    return out


def classify(op):
    if op.startswith("v_cmp"):
        return "valu_cmp"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "wait"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "vmem"


def main():
    obj = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "visual_odometry_amd/csrc/_obj/vo_score.o")
    needle = sys.argv[2] if len(sys.argv) > 2 else "score_kernelILi0E"
    anchor = sys.argv[3] if len(sys.argv) > 3 else "ds_read_b128"
    marker = sys.argv[4] if len(sys.argv) > 4 else "v_alignbit_b32"     # the loop must hold this opcode (the sign-bit tier 1)
    exclude = sys.argv[5] if len(sys.argv) > 5 else "v_min_f32"         # ... and not this one (the two-sided filter)
    ins = kernel_body(disassemble(obj), needle)
    index = {a: i for i, (a, _, _) in enumerate(ins)}
    best = None
    for i, (addr, op, text) in enumerate(ins):
        if not op.startswith("s_cbranch"):
            continue
        simm = int(text.split()[-1])
        if simm >= 32768:
            simm -= 65536
        target = addr + 4 + 4 * simm
        if target > addr or target not in index:
            continue
        body = ins[index[target]: i + 1]
        if any(o.startswith("s_cbranch") for _, o, _ in body[:-1]):
            continue                                   # not innermost
        fmas = sum(o.startswith("v_fma") for _, o, _ in body)
        if marker and not any(o.startswith(marker) for _, o, _ in body):
            continue
        if exclude and any(o.startswith(exclude) for _, o, _ in body):
            continue
        if best is None or fmas > best[0]:
            best = (fmas, body)
    if best is None:
        raise SystemExit("no inner loop found")
    body = best[1]
    per = sum(o == anchor for _, o, _ in body)
    mix = {}
    for _, o, _ in body:
        mix[classify(o)] = mix.get(classify(o), 0) + 1
    ops = {}
    for _, o, _ in body:
        ops[o] = ops.get(o, 0) + 1
    valu = mix.get("valu", 0) + mix.get("valu_cmp", 0)
    out = {
        "kernel": needle, "object": os.path.relpath(obj, ROOT), "loop_bytes": body[-1][0] + 4 - body[0][0],
        "evaluations_per_trip": per, "instructions_per_trip": len(body), "mix_per_trip": mix,
        "valu_per_evaluation": valu / per, "all_per_evaluation": len(body) / per,
        "non_valu_per_evaluation": (len(body) - valu) / per, "opcodes_per_trip": dict(sorted(ops.items())),
        "how": "scripts/isa_count.py: llvm-objdump of the gfx950 code object, innermost backward-branch loop with the most "
               "v_fma; one " + anchor + " per evaluation",
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
