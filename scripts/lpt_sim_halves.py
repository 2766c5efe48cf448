"""Simulate-first for VERDICT r5 item 2(i): what does cutting a quad's list into two jobs buy the backward blend's launch?

Input: a measured timeline of the 12 900 single-wave jobs (scripts/timeline.py -> gpurun_out/timeline_bwd.npz: start / end of every workgroup, the quad's
record counts). Model of a job: d = s + w, s = what a job costs before its first round and after its last (prologue trips, first gather, last flush: the
intercept of the measured length over the records it walked, and a second estimate from the shortest non-empty jobs), w = the rest. A quad that is cut at a round
boundary m becomes a back job [m, qdone) and a front job [0, m): each pays s (the front one + s_extra for the 16 bytes per pixel of state it starts from),
the work splits in proportion to the records. The jobs are list-scheduled per XCD over 384 wave slots in dispatch order — the same simulation as
scripts/lpt_sim.py, which reproduces the measured launch to ~3 %.

Variants: cut every quad with qdone >= thr (the review's proposal: thr = 192), cut only the quads of the last fraction of the dispatch order (what actually
forms the tail), both; halves dispatched in place or the fronts appended behind their backs.

    python scripts/lpt_sim_halves.py gpurun_out/timeline_bwd.npz [out.json]
"""
import heapq
import json
import sys

import numpy as np

d = np.load(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/timeline_bwd.npz")
t0, t1, qc, qd = d["t0"], d["t1"], d["qcount"], d["qdone"]
n = len(t0)
dur = t1 - t0
q, r = n >> 3, n & 7


def remap(b):
    x = b & 7
    base = x * (q + 1) if x < r else r * (q + 1) + (x - r) * q
    return base + (b >> 3)


job = np.array([remap(b) for b in range(n)])
rec = qd[job].astype(float)          # records the backward walks for block b
rec_all = qc[job].astype(float)
busy = dur[rec > 0]
A = np.stack([np.ones((rec > 0).sum()), rec[rec > 0]], 1)
coef, *_ = np.linalg.lstsq(A, busy, rcond=None)
s_fit, per_rec = float(coef[0]), float(coef[1])
short = np.sort(dur[(rec > 0) & (rec <= 64)])
s_short = float(np.median(short)) if len(short) else s_fit
out = {"jobs": int(n), "measured_launch_us": float(t1.max()), "mean_load_us": float(dur.sum() / 3072),
       "job_us_p10_p50_p90": [float(np.percentile(dur, p)) for p in (10, 50, 90)],
       "records_p10_p50_p90_max": [float(np.percentile(rec, p)) for p in (10, 50, 90, 100)],
       "fit_us": {"setup_intercept": s_fit, "per_record": per_rec, "corr": float(np.corrcoef(dur, rec)[0, 1]),
                  "median_of_one_round_jobs": s_short, "one_round_jobs": int(len(short))}}


def simulate(per_xcd_durs, slots=384):
    end = 0.0
    for durs in per_xcd_durs:
        h = [0.0] * slots
        heapq.heapify(h)
        for x in durs:
            heapq.heappush(h, heapq.heappop(h) + x)
        end = max(end, max(h))
    return end


by_xcd = [np.arange(n)[(np.arange(n) & 7) == x] for x in range(8)]
base = simulate([dur[b] for b in by_xcd])
out["simulated_dispatch_order_us"] = base


def variant(setup, s_extra, thr, last_frac, fronts_behind):
    per = []
    cut_jobs = 0
    for b in by_xcd:
        durs, tail = [], []
        first_cut = int(len(b) * (1.0 - last_frac))
        for i, blk in enumerate(b):
            dd = dur[blk]
            if rec[blk] >= thr and i >= first_cut and dd > 2 * setup:
                # cut at the round boundary nearest the middle (rounds of 64 records)
                rounds = int(np.ceil(rec[blk] / 64.0))
                mb = rounds // 2 * 64.0 / rec[blk] if rounds >= 2 else 0.5
                w = dd - setup
                back, front = setup + w * (1 - mb), setup + s_extra + w * mb
                cut_jobs += 1
                durs.append(back)
                (tail if fronts_behind else durs).append(front)
            else:
                durs.append(dd)
        per.append(durs + tail)
    return simulate(per), cut_jobs


res = []
# (the fit extrapolates from 230-350 records down to zero: the sweep below also asks what the cut would buy if a job's fixed cost were much smaller than the intercept)
for setup_name, setup in (("intercept", max(s_fit, 0.0)), ("assume-8us", 8.0), ("assume-4us", 4.0)):
    for s_extra in (0.0, 1.0):
        for thr in (128, 192, 256, 320):
            for last_frac in (1.0, 0.5, 0.3, 0.2):
                for fronts_behind in (False, True):
                    end, cut = variant(setup, s_extra, thr, last_frac, fronts_behind)
                    res.append({"setup": setup_name, "setup_us": setup, "front_extra_us": s_extra, "min_records": thr, "last_fraction_of_dispatch_order": last_frac,
                                "fronts_behind": fronts_behind, "jobs_cut": cut, "launch_us": end, "gain_us": base - end})
res.sort(key=lambda x: -x["gain_us"])
out["best_10"] = res[:10]
out["review_proposal_thr192_all"] = [x for x in res if x["min_records"] == 192 and x["last_fraction_of_dispatch_order"] == 1.0]
out["best_by_setup_assumption"] = {k: max((x for x in res if x["setup"] == k), key=lambda x: x["gain_us"]) for k in ("intercept", "assume-8us", "assume-4us")}
best_measured = out["best_by_setup_assumption"]["intercept"]["gain_us"]
out["verdict"] = "build" if best_measured >= 12.0 else "kill: %.1f us projected with the measured fixed cost per job (criterion: >= 12 us)" % best_measured
print(json.dumps(out, indent=1))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
