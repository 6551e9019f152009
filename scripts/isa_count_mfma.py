#!/usr/bin/env python3
"""Instruction mix of the first tier of the F count kernel (score_fund_mfma_kernel), read from the code object that ships
in libvo_hip.so.   usage: isa_count_mfma.py [object file] > profiles/r3_score_fund_isa.json

The tier is straight-line code per group of 64 correspondences x 64 models per wave: 4 tiles of 32 x 32 evaluations, each
3 v_mfma_f32_32x32x16_f16 and an epilogue of one v_fma_f32 + one v_alignbit_b32 per accumulator pair.  Counted: everything
from the group's first MFMA through the last v_alignbit_b32 that follows its 12th MFMA; one lane holds 64 of the group's
4096 evaluations, so "per evaluation" = per trip / 64.  bench.py prices the kernel against the MFMA peak (useful
multiply-adds of the split inner products) and, beside it, against the VALU issue rate with the figure found here."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from isa_count import ROOT, classify, disassemble, kernel_body  # noqa: E402


def main():
    obj = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "visual_odometry_amd/csrc/_obj/vo_score.o")
    ins = kernel_body(disassemble(obj), "score_fund_mfma_kernel")
    mf = [i for i, (_, o, _) in enumerate(ins) if o.startswith("v_mfma")]
    if len(mf) != 12:
        raise SystemExit(f"expected the 12 MFMAs of one group, found {len(mf)}")
    end = mf[-1]
    last = end
    for i in range(end, len(ins)):
        if ins[i][1].startswith("s_cbranch") or ins[i][1].startswith("s_branch"):
            break
        if ins[i][1] == "v_alignbit_b32":
            last = i
    body = ins[mf[0]: last + 1]
    mix, ops = {}, {}
    for _, o, _ in body:
        c = "mfma" if o.startswith("v_mfma") else classify(o)
        mix[c] = mix.get(c, 0) + 1
        ops[o] = ops.get(o, 0) + 1
    valu = mix.get("valu", 0) + mix.get("valu_cmp", 0)
    per = 64
    k_used = {"d": 8 * 3 + 2, "den2": 5 * 3 + 1}
    out = {
        "kernel": "score_fund_mfma_kernel", "object": os.path.relpath(obj, ROOT), "evaluations_per_trip_per_lane": per,
        "evaluations_per_trip_per_wave": 4096, "instructions_per_trip": len(body), "mix_per_trip": mix,
        "valu_per_evaluation": valu / per, "all_per_evaluation": len(body) / per,
        "mfma_per_1024_evaluations": mix.get("mfma", 0) / 4.0,
        "mfma_flop_per_evaluation_issued": mix.get("mfma", 0) * 2 * 32 * 32 * 16 / 4096.0,
        "mfma_flop_per_evaluation_useful": 2 * (k_used["d"] + k_used["den2"]),
        "k_entries_used": k_used, "opcodes_per_trip": dict(sorted(ops.items())),
        "how": "scripts/isa_count_mfma.py: llvm-objdump of the gfx950 code object, the straight-line block from the first of the "
               "group's 12 v_mfma through the last v_alignbit_b32 of its fourth epilogue",
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
