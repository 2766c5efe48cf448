"""Turns gpurun_out/prof_<tag>/ (scripts/profile_round.sh) into the committed summaries under profiles/:
   <tag>_kernel_stats.md, <tag>_pmc.md, <tag>_bench.json and rNN_traffic.json (read by bench.py)."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01x"
note = sys.argv[2] if len(sys.argv) > 2 else ""
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")


def short(name):
    name = name.split("(")[0].replace("void ", "")
    if "gsr::" not in name:
        return name.split("<")[0]
    name = name.replace("gsr::", "")
    if name.startswith("K_blend"):   # K_blend_fwd<64, DUAL>, K_blend_bwd<64, DUAL, COLORS>
        args = [x.strip() for x in name[name.index("<") + 1:name.rindex(">")].split(",")] if "<" in name else []
        dual = len(args) > 1 and args[1] == "true"
        nocol = len(args) > 2 and args[2] == "false"
        return name.split("<")[0] + ("_dual" if dual else "") + ("_nocolour" if nocol else "")
    return name.split("<")[0]


bench = None
for line in open(os.path.join(src, "bench.json")):
    if line.startswith("{"):
        bench = json.loads(line)
json.dump(bench, open(os.path.join(dst, tag + "_bench.json"), "w"), indent=1)

# ---- kernel stats
rows = []
for f in glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
with open(os.path.join(dst, tag + "_kernel_stats.md"), "w") as o:
    o.write("# Round %s, profile %s — %s\n\n" % (tag[1:3].lstrip("0"), tag, note))
    o.write("Command (MI355X box, scripts/profile_round.sh): `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --mode rasterize --steps 20 --warmup 5 --no-cpu`\n\n")
    c = bench["config"]
    o.write("Workload: %d Gaussians, %dx%d, V = %d visible, R = %d tile instances; one step = gsr_forward_ws + gsr_backward.\n\n"
            % (c["splats"], c["width"], c["height"], c["visible"], c["tile_instances"]))
    r = bench["roofline"]
    o.write("Un-profiled bench.py of the same build (profiles/%s_bench.json): %.3f ms/step; live HIP-event averages: K_blend_bwd %.1f us, K_blend_fwd %.1f us.\n\n"
            % (tag, bench["ms_per_step"], r["avg_launch_ms"] * 1e3, r["fwd_blend_avg_launch_ms"] * 1e3))
    o.write("| kernel | calls | avg us | min us | max us | % of GPU time |\n|---|---|---|---|---|---|\n")
    for x in rows:
        if float(x["Percentage"]) < 0.1:
            continue
        o.write("| %s | %s | %.1f | %.1f | %.1f | %s |\n" % (short(x["Name"]), x["Calls"], float(x["AverageNs"]) / 1e3,
                                                         float(x["MinNs"]) / 1e3, float(x["MaxNs"]) / 1e3, x["Percentage"]))

# ---- per-kernel resources of the build (VERDICT r4 hygiene item: occupancy claims belong beside the timings)
try:
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import kres
    res = kres.resources()
    with open(os.path.join(dst, tag + "_kernel_stats.md"), "a") as o:
        o.write("\n## Kernel resources of this build (scripts/kres.py: metadata of the gfx950 code object)\n\n")
        o.write("| kernel | VGPRs (unified file: 512 per SIMD lane) | SGPRs | LDS bytes | scratch bytes | spilled VGPRs | waves per SIMD the registers allow | single-wave or workgroup waves per CU the LDS allows |\n|---|---|---|---|---|---|---|---|\n")
        for r in res:
            if not r["name"].startswith(("K_preprocess", "K_bin", "K_tile_sort_cut", "K_blend", "K_splat_bwd", "K_map", "K_ssim", "K_track", "K_pose", "K_composite", "K_shard")):
                continue
            alloc = max(8, -(-r["vgpr"] // 8) * 8)
            lds_blocks = -(-r["lds"] // 1280) if r["lds"] else 0
            o.write("| `%s` | %d | %d | %d | %d | %d | %d | %s |\n" % (r["name"], r["vgpr"], r["sgpr"], r["lds"], r["scratch"], r["spill"], min(8, 512 // alloc),
                                                                  ("%d workgroups" % (128 // lds_blocks)) if lds_blocks else "-"))
except Exception as e:   # (no hipcc: the table is skipped, the timings stand)
    print("kernel resources skipped:", e)

# ---- PMC
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(os.path.join(src, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = short(row["Kernel_Name"])
        if not k.startswith("K_"):
            continue
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[k][row["Counter_Name"]] += 1
avg = {k: {c: v / cnt[k][c] for c, v in d.items()} for k, d in agg.items()}
# durations of the kernels in the SAME pass that collected GRBM_GUI_ACTIVE (pmc_condense.py writes them): the clock and everything per SIMD derived from it
pass_us = {}
for f in glob.glob(os.path.join(src, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    if not any(r_["Counter_Name"] == "GRBM_GUI_ACTIVE" for r_ in csv.DictReader(open(f))):
        continue
    for g_ in glob.glob(os.path.join(os.path.dirname(f), "pass_kernel_durations.csv")):
        tot, n_ = collections.defaultdict(float), collections.defaultdict(int)
        for row in csv.DictReader(open(g_)):
            k = short(row["Kernel_Name"]); tot[k] += float(row["AverageNs"]) * float(row["Launches"]); n_[k] += float(row["Launches"])
        pass_us = {k: tot[k] / n_[k] / 1e3 for k in tot}
order = ["K_preprocess", "K_bin_count", "K_bin_colscan", "K_scan_tiles", "K_bin_fill", "K_tile_sort_cut", "K_tile_sort_short", "K_tile_sort_long", "K_blend_fwd", "K_blend_bwd", "K_splat_bwd"]
sq = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY",
      "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_WAIT_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE",
      "SQ_BUSY_CU_CYCLES", "GRBM_GUI_ACTIVE"]
stats_avg = {short(x["Name"]): float(x["AverageNs"]) / 1e3 for x in rows}
with open(os.path.join(dst, tag + "_pmc.md"), "w") as o:
    o.write("# Round %s, profile %s — PMC counters (rocprofv3 --pmc, separate passes, --kernel-trace only)\n\n" % (tag[1:3].lstrip("0"), tag))
    o.write("Per launch, averaged over the launches of `bench.py --no-cpu --mode rasterize --steps 3 --warmup 1`. GRBM_GUI_ACTIVE is summed over the 8 XCDs (/8 = shader cycles of the launch). SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are in quad-cycles (x4 = shader cycles).\n\n")
    o.write("| kernel | " + " | ".join(sq) + " |\n|---|" + "---|" * len(sq) + "\n")
    for k in order:
        if k in avg:
            o.write("| %s | " % k + " | ".join("%.3g" % avg[k].get(c, float("nan")) for c in sq) + " |\n")
    o.write("\nDerived (per launch). The clock and the per-SIMD figures divide this pass's GRBM_GUI_ACTIVE by the kernel's duration IN THE SAME PASS (counter passes serialise and slow "
            "the launches; round 5 divided by the --stats pass's duration and printed up to 4.36 GHz on a 2.4 GHz part); 'us' is the rocprofv3 --stats average, 'us in the counter pass' what the clock uses:\n\n")
    o.write("| kernel | us | clock GHz (GUI_ACTIVE/8/us) | waves resident per SIMD (WAVE_CYCLES*4/cycles/1024) | VALU insts / us (chip, thousands) | cycles per VALU inst per wave | wave time: active / issue-stalled / waiting | LDS pipe busy (LDS_IDX_ACTIVE / BUSY_CU_CYCLES) |\n|---|---|---|---|---|---|---|---|\n")
    for k in order:
        if k in avg and k in stats_avg and "SQ_WAVE_CYCLES" in avg[k]:
            a_ = avg[k]
            us = stats_avg[k]
            cyc = a_.get("GRBM_GUI_ACTIVE", float("nan")) / 8
            wc = a_["SQ_WAVE_CYCLES"] * 4
            pus = pass_us.get(k, us)
            # GRBM_GUI_ACTIVE spans the dispatch and the drain of a launch as well: for a kernel of a few tens of microseconds it is 1.2-1.8 x the kernel's own
            # cycles (K_bin_colscan: 17 us of "GUI active" around a 9.9 us kernel) — as a clock that reads 2.8-4.2 GHz on a 2.4 GHz part. The cycles a kernel
            # can have had are capped at its duration x 2.4 GHz; the clock column says which kernels the cap applied to.
            capped = cyc > pus * 2400.0
            cyc = min(cyc, pus * 2400.0)
            o.write("| %s | %.1f (%.1f in the counter pass) | %s | %.2f | %.2f | %.1f | %.0f %% / %.0f %% / %.0f %% | %.2f |\n" % (
                k, us, pus, ("2.40 (cap: GUI_ACTIVE window %.1fx the kernel)" % (a_.get("GRBM_GUI_ACTIVE", 0) / 8 / (pus * 2400.0))) if capped else "%.2f" % (cyc / pus / 1e3),
                wc / cyc / 1024 if cyc == cyc else float("nan"),
                a_["SQ_INSTS_VALU"] / us / 1e3, wc / max(a_["SQ_INSTS_VALU"], 1),
                100 * a_.get("SQ_ACTIVE_INST_ANY", float("nan")) / a_["SQ_WAVE_CYCLES"], 100 * a_["SQ_WAIT_INST_ANY"] / a_["SQ_WAVE_CYCLES"],
                100 * a_["SQ_WAIT_ANY"] / a_["SQ_WAVE_CYCLES"],
                a_["SQ_LDS_IDX_ACTIVE"] / a_["SQ_BUSY_CU_CYCLES"] if a_.get("SQ_BUSY_CU_CYCLES") and a_.get("SQ_LDS_IDX_ACTIVE") else 0.0))
    o.write("\nVALU instruction classes on gfx950 (scripts/valu_bench.hip, valu_bench2.hip, wave-instructions/s per SIMD at 8 waves): "
            "v_add/mul/fma/and/mov 0.9-1.0 G (2 cycles); v_cmp, v_cndmask, v_min/max, shifts, v_mad_u32_u24, every DPP op, v_pk_* 0.5-0.58 G "
            "(4 cycles); v_exp/rcp/log/sqrt, v_permlane*_swap 0.29 G (8 cycles); f32 MFMA 16x16x4 0.07 G (32 cycles, no overlap with VALU).\n")
    o.write("\n## L2 <-> fabric traffic\n\nFETCH_SIZE / WRITE_SIZE are reported in KB. Per MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 reports 1/2 of the bytes of a streaming read; traffic = 2*FETCH_SIZE + WRITE_SIZE.\n\n")
    o.write("| kernel | FETCH_SIZE KB | WRITE_SIZE KB | traffic MB (2F+W) | TCC hit rate |\n|---|---|---|---|---|\n")
    tj = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --steps 3 --warmup 1 --no-cpu; traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024 bytes per launch",
          "profile": tag, "workload": "1M splats 1200x680", "kernels": {}}
    for k in order:
        if k in avg and "FETCH_SIZE" in avg[k]:
            F, Wr = avg[k]["FETCH_SIZE"], avg[k].get("WRITE_SIZE", 0.0)
            hit, miss = avg[k].get("TCC_HIT_sum", float("nan")), avg[k].get("TCC_MISS_sum", float("nan"))
            tr = (2 * F + Wr) * 1024
            o.write("| %s | %.0f | %.0f | %.1f | %.2f |\n" % (k, F, Wr, tr / 1e6, hit / (hit + miss) if hit == hit else float("nan")))
            tj["kernels"][k] = {"fetch_size_kb": F, "write_size_kb": Wr, "traffic_bytes": tr}
    for k in ("K_blend_fwd", "K_blend_bwd"):   # what bench.py's roofline_valu quotes
        if k in avg and k in stats_avg and k in tj["kernels"]:
            tj["kernels"][k]["valu_instructions"] = avg[k]["SQ_INSTS_VALU"]
            tj["kernels"][k]["profiled_launch_us"] = stats_avg[k]
            if avg[k].get("SQ_ACTIVE_INST_LDS"):
                tj["kernels"][k]["lds_conflict_frac"] = avg[k]["SQ_LDS_BANK_CONFLICT"] / avg[k]["SQ_ACTIVE_INST_LDS"]
            if avg[k].get("SQ_BUSY_CU_CYCLES") and avg[k].get("SQ_LDS_IDX_ACTIVE"):
                tj["kernels"][k]["lds_pipe_busy"] = avg[k]["SQ_LDS_IDX_ACTIVE"] / avg[k]["SQ_BUSY_CU_CYCLES"]
            if avg[k].get("GRBM_GUI_ACTIVE"):
                tj["kernels"][k]["clock_ghz"] = min(2.4, avg[k]["GRBM_GUI_ACTIVE"] / 8 / pass_us.get(k, stats_avg[k]) / 1e3)
            if avg[k].get("SQ_WAVE_CYCLES") and avg[k].get("GRBM_GUI_ACTIVE"):
                tj["kernels"][k]["waves_per_simd"] = avg[k]["SQ_WAVE_CYCLES"] * 4 / min(avg[k]["GRBM_GUI_ACTIVE"] / 8, pass_us.get(k, stats_avg[k]) * 2400.0) / 1024
    json.dump(tj, open(os.path.join(dst, tag[:3] + "_traffic.json"), "w"), indent=1)

# ---- per-kernel roofline table: algorithmic bytes (SURVEY.md 8d per stage, DESIGN.md section 4's table) / rocprof average / fraction of 8 TB/s / counter traffic
c = bench["config"]
P_, V_, R_, N_ = c["splats"], c["visible"], c["tile_instances"], c["width"] * c["height"]
T_ = ((c["width"] + 15) // 16) * ((c["height"] + 15) // 16)
rows_ = min(512, -(-P_ // 4096))
alg = {"K_preprocess": ("20P + 72V", 20 * P_ + 72 * V_), "K_bin_count": ("16P (records read once) + 4 rows T (its row of the count matrix)", 16 * P_ + 4 * rows_ * T_),
       "K_bin_colscan": ("8 rows T (count matrix in, offsets out) + 16T", 8 * rows_ * T_ + 16 * T_), "K_bin_fill": ("16P + 8R", 16 * P_ + 8 * R_),
       "K_tile_sort_cut": ("12R (keys in, list out) + 8R (reach gather) + 8 x quad hits (~1.35 R)", int(20 * R_ + 8 * 1.35 * R_)),
       "K_blend_fwd": ("44R + 24N", 44 * R_ + 24 * N_), "K_blend_bwd": ("40R + 20N + 36V", 40 * R_ + 20 * N_ + 36 * V_), "K_splat_bwd": ("8P + 208V", 8 * P_ + 208 * V_)}
with open(os.path.join(dst, tag + "_kernel_stats.md"), "a") as o:
    o.write("\n## Every kernel against the HBM roofline (8 TB/s): algorithmic bytes / rocprof average / fraction, and the counters' traffic beside it\n\n")
    o.write("P = %d, V = %d, R = %d, N = %d pixels, T = %d tiles, %d splat ranges. Traffic = 2 x FETCH_SIZE + WRITE_SIZE of the PMC passes (profiles/%s_pmc.md).\n\n" % (P_, V_, R_, N_, T_, rows_, tag))
    o.write("| kernel | algorithmic bytes (formula) | MB | avg us | GB/s | fraction of 8 TB/s | traffic MB | traffic / algorithmic |\n|---|---|---|---|---|---|---|---|\n")
    tot_b = tot_us = 0.0
    for k in order:
        if k in alg and k in stats_avg:
            fm, b_ = alg[k]
            us = stats_avg[k]
            tr = tj["kernels"].get(k, {}).get("traffic_bytes")
            tot_b += b_; tot_us += us
            o.write("| %s | %s | %.1f | %.1f | %.0f | %.3f | %s | %s |\n" % (k, fm, b_ / 1e6, us, b_ / us / 1e3, b_ / us / 1e3 / 8000.0,
                                                                          "%.1f" % (tr / 1e6) if tr else "-", "%.2f" % (tr / b_) if tr else "-"))
    o.write("| sum | | %.1f | %.1f | %.0f | %.3f | | |\n" % (tot_b / 1e6, tot_us, tot_b / tot_us / 1e3, tot_b / tot_us / 1e3 / 8000.0))
print("written", tag)
