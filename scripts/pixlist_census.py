"""Second census of the backward blend (round 4): what the forward could tell the backward — blended pixels per patch hit, and the
loop lengths with (Z) lists that drop the patch hits blending no pixel, (P) per-pixel lists inside 16-entry windows.
oracle/gsr_oracle.c:gsro_pixlist_census.   python scripts/pixlist_census.py [out.json] [name filter]"""
import json, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R + '/tests')
from conftest import load_package
from oracle import oracle
gsr = load_package(); syn = gsr.synthetic
out = {}
for name, P, cam, mult in (("headline 1M replica", 1_000_000, syn.REPLICA, 1.0), ("fat x4 1M replica", 1_000_000, syn.REPLICA, 4.0),
                           ("scannet 2M", 2_000_000, syn.CAMERAS["scannet"], 1.0)):
    if len(sys.argv) > 2 and sys.argv[2] not in name:
        continue
    t0 = time.time()
    c = syn.make_camera(**cam); sc = syn.make_scene(P, c, seed=0, scale_mult=mult)
    o = oracle.Oracle(omp=True)
    o.forward(copy_stages=False, means3D=sc.means3D, opacities=sc.opacities, cam=sc.cam, colors=sc.colors, scales=sc.scales, rotations=sc.rotations)
    a = o.pixlist_census()
    hits = sum(a[:17]); pairs = sum(i * a[i] for i in range(17))
    d = dict(patch_hits=hits, blended_pairs=pairs, patch_hits_by_blended_pixels=a[:17], patch_hits_blending_nothing=a[0] / hits,
             quad_hits=a[17], quad_hits_blending_nothing=a[18] / max(a[17], 1), rounds=a[25],
             today=dict(wave_iterations=a[19], reduce_phases=a[20]),
             Z=dict(wave_iterations=a[21], reduce_phases=a[22]),
             P=dict(wave_iterations=a[23], windows=a[24]), PZ=dict(wave_iterations=a[26]), P_round=dict(wave_iterations=a[27]))
    out[name] = d
    print(name, json.dumps(d), "(%.0f s)" % (time.time() - t0), flush=True)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
