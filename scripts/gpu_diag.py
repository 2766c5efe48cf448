import sys, os, numpy as np, torch
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,R+'/tests')
from conftest import load_package
gsr=load_package(); syn=gsr.synthetic
from oracle import oracle
from util import rel_err, run_oracle
P=int(sys.argv[1]) if len(sys.argv)>1 else 300000
cam=syn.make_camera(**syn.REPLICA)
sc=syn.make_scene(P,cam,seed=0)
o,f,b=run_oracle(sc,oracle,backward=False)
s=gsr.capi.Settings.from_camera(cam)
st=gsr.forward(s,sc.means3D,sc.opacities,colors=sc.colors,scales=sc.scales,rotations=sc.rotations)
torch.cuda.synchronize()
col=st.color.cpu().numpy()
diff=np.abs(col-f.color).max(0)
print(os.environ.get('GSR_LIB_OVERRIDE'),'color rel',rel_err(col,f.color),'n bad px (>1e-5)',(diff>1e-5).sum(),'of',diff.size)
ys,xs=np.nonzero(diff>1e-5)
rng=f.stages['ranges']; cnt=(rng[:,1]-rng[:,0]).astype(int)
gx=75
tiles=(ys//16)*gx+(xs//16)
ut,uc=np.unique(tiles,return_counts=True)
print('bad tiles',len(ut),'their list sizes min/med/max',cnt[ut].min() if len(ut) else None, np.median(cnt[ut]) if len(ut) else None, cnt[ut].max() if len(ut) else None,'all tiles max',cnt.max(), 'tiles>256:',(cnt>256).sum())
for t,c in list(zip(ut,uc))[:10]: print(' tile',t,'n',cnt[t],'badpx',c)
if len(ys):
    i=np.argmax(diff[ys,xs]); y,x=ys[i],xs[i]; t=tiles[i]
    print('worst px',x,y,'diff',diff[y,x],'gpu',col[:,y,x],'ora',f.color[:,y,x],'ncontrib',f.stages['n_contrib'][y*1200+x], 'lx,ly',x%16,y%16)
