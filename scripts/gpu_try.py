import sys, time, numpy as np, torch
import os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,R+'/tests')
from conftest import load_package
gsr=load_package(); syn=gsr.synthetic
from oracle import oracle
from util import rel_err, pose, run_oracle
def run(P,camkw,mode='rgb',mult=1.0,Tcw=None,bg=(0,0,0),seed=0,**kw):
    cam=syn.make_camera(**camkw,Tcw=Tcw,bg=bg)
    sc=syn.make_scene(P,cam,seed=seed,scale_mult=mult,color_mode=mode,**kw)
    o,f,b=run_oracle(sc,oracle)
    s=gsr.capi.Settings.from_camera(cam)
    st=gsr.forward(s,sc.means3D,sc.opacities,colors=sc.colors,shs=sc.shs,scales=sc.scales,rotations=sc.rotations)
    torch.cuda.synchronize()
    d=gsr.debug_export(st)
    print(f"P={P} {mode} x{mult} R={st.num_rendered} vs {f.num_rendered}")
    print("  radii eq",np.array_equal(st.radii.cpu().numpy(),f.radii),"tiles eq",np.array_equal(d['tiles_touched'],f.stages['tiles_touched']),
          "plist eq",np.array_equal(d['point_list'],f.stages['point_list']),"ranges eq",np.array_equal(d['ranges'],f.stages['ranges']),
          "keys eq",np.array_equal(d['point_list_keys'],f.stages['keys_sorted']))
    print("  means2D exact",np.array_equal(d['means2D'],f.stages['means2D']),"conic exact",np.array_equal(d['conic_opacity'],f.stages['conic_opacity']), "depths exact",np.array_equal(d['depths'],f.stages['depths']))
    col=st.color.cpu().numpy(); dep=st.depth.cpu().numpy()
    print("  color rel",rel_err(col,f.color),"depth mismatch frac",(dep!=f.depth).mean(),"ncontrib mismatch",(d['n_contrib']!=f.stages['n_contrib']).mean(),"finalT rel",rel_err(d['final_T'],f.stages['final_T']))
    g=gsr.backward(st,sc.dL_dpix); torch.cuda.synchronize()
    for n in ['dL_dmeans2D','dL_dconic','dL_dopacity','dL_dcolors','dL_dmeans3D','dL_dcov3D','dL_dsh','dL_dscales','dL_drotations']:
        print("   ",n,rel_err(getattr(g,n).cpu().numpy(),getattr(b,n)))
run(2000,dict(width=160,height=120,fx=120.,fy=118.))
run(10000,syn.TUM1)
run(10000,syn.TUM1,mode='depth',mult=4.0)
run(3000,dict(width=200,height=150,fx=150.,fy=150.),mode='sh',mult=3.0,Tcw=pose(),bg=(0.3,0.5,0.7),frac_behind=0.1,frac_offscreen=0.3)
run(300000,syn.REPLICA)
