#!/usr/bin/env python3
"""Per-kernel resources of a build (VGPRs, AGPRs, SGPRs, LDS bytes, scratch, spills): compiles csrc/gsr_api.hip with -save-temps
and reads the metadata of the gfx950 assembly. `python scripts/kres.py [-Dflags...] [filter]`; with `--md` a markdown table."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gsorb-slam_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared"]

def resources(defs=(), out_so=None):
    with tempfile.TemporaryDirectory() as d:
        so = out_so or os.path.join(d, "x.so")
        subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *defs, "-save-temps=obj", "-o", so, os.path.join(CSRC, "gsr_api.hip")], check=True, cwd=d)
        base = os.path.dirname(so)
        cand = [os.path.join(p, f) for p in {d, base} for f in os.listdir(p) if f.endswith("gfx950.s")]
        txt = open(cand[0]).read()
        for p in {d, base}:
            for f in os.listdir(p):
                if f.startswith("gsr_api-") and not f.endswith(".so"):
                    os.remove(os.path.join(p, f))
    res = []
    for blk in txt.split("  - .agpr_count:")[1:]:
        g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, blk).group(1)
        name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name.replace("void ", "").replace("gsr::", ""))
        res.append(dict(name=name, vgpr=int(g("vgpr_count")), agpr=int(blk.split()[0]), sgpr=int(g("sgpr_count")), lds=int(g("group_segment_fixed_size")),
                        scratch=int(g("private_segment_fixed_size")), spill=int(g("vgpr_spill_count"))))
    return res

if __name__ == "__main__":
    args = sys.argv[1:]
    md = "--md" in args
    defs = [a for a in args if a.startswith("-D")]
    out = [a[2:] for a in args if a.startswith("-o")]
    filt = [a for a in args if not a.startswith("-")]
    rows = [r for r in resources(defs, out[0] if out else None) if not filt or any(f in r["name"] for f in filt)]
    if md:
        print("| kernel | VGPR | AGPR | SGPR | LDS B | scratch B | spilled VGPRs | waves/SIMD (registers) |\n|---|---|---|---|---|---|---|---|")
    for r in rows:
        tot = r["vgpr"]  # on gfx950 .vgpr_count is the unified total
        waves = min(8, 512 // max(8, -(-tot // 8) * 8))
        if md:
            print("| `%s` | %d | %d | %d | %d | %d | %d | %d |" % (r["name"], r["vgpr"], r["agpr"], r["sgpr"], r["lds"], r["scratch"], r["spill"], waves))
        else:
            print("%-52s vgpr %3d agpr %3d sgpr %3d lds %6d scratch %4d spill %3d" % (r["name"][:52], r["vgpr"], r["agpr"], r["sgpr"], r["lds"], r["scratch"], r["spill"]))
