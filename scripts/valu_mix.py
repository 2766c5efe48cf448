"""Static VALU class mix of the two blend kernels (what bench.py's roofline_valu charges per instruction instead of a flat 4
cycles): disassembles gsr_api.hip for gfx950 (hipcc -S) and sorts every VALU instruction of K_blend_fwd / K_blend_bwd into
the cost classes measured by scripts/valu_bench.hip / valu_bench2.hip (2 / 4 / 8 cycles per wave instruction).
STATIC counts — every instruction once, whatever its trip count; the hot loops have about the same mix.
    python scripts/valu_mix.py > profiles/r03_valu_mix.json"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gsorb-slam_amd", "csrc")
FULL = ("v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32",
        "v_or_b32", "v_xor_b32", "v_mov_b32", "v_add_co_u32", "v_addc_co_u32", "v_mul_legacy_f32", "v_add3_u32", "v_lshl_add_u32", "v_and_or_b32",
        "v_or3_b32", "v_lshl_or_b32", "v_add_lshl_u32", "v_accvgpr")
QUARTER = ("v_exp_f32", "v_rcp_f32", "v_log_f32", "v_sqrt_f32", "v_rsq_f32", "v_rcp_iflag_f32", "v_permlane", "v_mul_lo_u32", "v_mul_hi_u32",
           "v_mad_u64_u32", "v_mad_i64_i32")


def cls(op, text):
    if "dpp" in op or " row_" in text or "quad_perm" in text:
        return "half"
    if op.startswith(QUARTER):
        return "quarter"
    if op.startswith(FULL):
        return "full"
    return "half"     # compares, selects, min/max/med3, shifts, bfe, mads, conversions, packed ops, readlane ...


def main():
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-S",
                        "--cuda-device-only", "-o", asm, os.path.join(CSRC, "gsr_api.hip")], check=True, cwd=CSRC, stderr=subprocess.DEVNULL)
        text = open(asm).read()
    out = {"source": "scripts/valu_mix.py: static instruction mix (hipcc -S for gfx950), cycles per class from scripts/valu_bench*.hip",
           "cycles": {"full": 2, "half": 4, "quarter": 8}, "kernels": {}}
    for name, pat in (("K_blend_fwd", r"^_ZN3gsr11K_blend_fwdILi64ELb0E[^:\n]*:"), ("K_blend_bwd", r"^_ZN3gsr11K_blend_bwdILi64ELb0E[^:\n]*:")):
        m = re.search(pat, text, re.M)
        body = text[m.end():text.index("s_endpgm", m.end())]
        n = {"full": 0, "half": 0, "quarter": 0}
        for line in body.splitlines():
            t = line.strip()
            if t.startswith("v_") and not t.startswith("v_nop"):
                n[cls(t.split()[0], t)] += 1
        tot = sum(n.values())
        out["kernels"][name] = {"static_valu_instructions": tot, **n,
                                "cycles_per_instruction": (2 * n["full"] + 4 * n["half"] + 8 * n["quarter"]) / max(tot, 1)}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
