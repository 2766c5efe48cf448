"""Offline list-scheduling simulation of the backward blend's 12 900 single-wave jobs over the chip's wave slots, from a measured
timeline (scripts/timeline.py -> gpurun_out/timeline_bwd.npz): what the launch would take if the jobs were dispatched longest-first
(by their true length, or by a cost proxy the pipeline knows before the launch: the quad's record count)."""
import heapq, sys
import numpy as np
d = np.load(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/timeline_bwd.npz")
t0, t1, xcc, qc, qd = d["t0"], d["t1"], d["xcc"], d["qcount"], d["qdone"]
n = len(t0); dur = t1 - t0
# block b -> job w (xcd_remap): xcd = b & 7, w = base(xcd) + (b >> 3)
nb = n; q, r = nb >> 3, nb & 7
def remap(b):
    x = b & 7
    base = x * (q + 1) if x < r else r * (q + 1) + (x - r) * q
    return base + (b >> 3)
job = np.array([remap(b) for b in range(n)])          # quad index (tile*4+quad) of block b
cost_qd = qd[job].astype(float); cost_qc = qc[job].astype(float)
print("corr(duration, qdone) %.3f  corr(duration, qcount) %.3f" % (np.corrcoef(dur, cost_qd)[0, 1], np.corrcoef(dur, cost_qc)[0, 1]))
def simulate(order_per_xcd, slots_per_xcd=384, launch_gap=0.0):
    end = 0.0
    for x in range(8):
        blocks = order_per_xcd[x]
        h = [0.0] * slots_per_xcd; heapq.heapify(h)
        for b in blocks:
            t = heapq.heappop(h); heapq.heappush(h, t + dur[b] + launch_gap)
        end = max(end, max(h))
    return end
by_xcd = [np.arange(n)[(np.arange(n) & 7) == x] for x in range(8)]
print("measured launch %.1f us, mean load %.1f us" % (t1.max(), dur.sum() / 3072))
print("simulated, dispatch order       : %.1f us" % simulate(by_xcd))
print("simulated, LPT by true duration : %.1f us" % simulate([b[np.argsort(-dur[b])] for b in by_xcd]))
print("simulated, LPT by qdone         : %.1f us" % simulate([b[np.argsort(-cost_qd[b], kind='stable')] for b in by_xcd]))
print("simulated, LPT by qcount        : %.1f us" % simulate([b[np.argsort(-cost_qc[b], kind='stable')] for b in by_xcd]))
for nbk in (8, 16, 32, 64):   # bucketed LPT: linear buckets between the XCD's min and max cost
    o = []
    for b in by_xcd:
        c = cost_qd[b]; k = np.floor((c - c.min()) / max(c.max() - c.min(), 1) * (nbk - 1e-6)).astype(int)
        o.append(b[np.argsort(-k, kind='stable')])
    print("simulated, %2d linear buckets of qdone: %.1f us" % (nbk, simulate(o)))
for bits in (2, 3, 4):        # log-scale buckets: 2^bits per octave
    o = []
    for b in by_xcd:
        c = np.maximum(cost_qd[b], 1); k = np.floor(np.log2(c) * (1 << bits)).astype(int)
        o.append(b[np.argsort(-k, kind='stable')])
    print("simulated, log buckets %2d per octave of qdone: %.1f us" % (1 << bits, simulate(o)))
# shortest-last only: the last `tail` jobs of each XCD are its shortest ones (everything else in dispatch order)
for frac in (0.1, 0.2, 0.3):
    o = []
    for b in by_xcd:
        m = int(len(b) * frac); idx = np.argsort(cost_qd[b], kind='stable')[:m]; mask = np.ones(len(b), bool); mask[idx] = False
        o.append(np.concatenate([b[mask], b[idx][np.argsort(-cost_qd[b][idx], kind='stable')]]))
    print("simulated, shortest %.0f %% last: %.1f us" % (frac * 100, simulate(o)))
