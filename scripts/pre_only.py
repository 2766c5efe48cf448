"""K_preprocess alone, timed between HIP events over N back-to-back forwards' preprocess stage (GSR_LIB_OVERRIDE picks the library)."""
import sys, os, numpy as np, torch, ctypes as C
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,R+'/tests')
from conftest import load_package
gsr=load_package(); syn=gsr.synthetic
cam=syn.make_camera(**syn.REPLICA); sc=syn.make_scene(1000000,cam,seed=0)
s=gsr.capi.Settings.from_camera(cam)
t=lambda x: torch.as_tensor(x,dtype=torch.float32,device='cuda').contiguous()
ins=dict(means3D=t(sc.means3D),opacities=t(sc.opacities),colors=t(sc.colors),shs=None,scales=t(sc.scales),rotations=t(sc.rotations),cov3D=None)
ws=gsr.capi.Workspace(1000000,1200,680,max_rendered=4000000)
hip=C.CDLL("libamdhip64.so.7"); hip.hipEventCreate.argtypes=[C.POINTER(C.c_void_p)]; hip.hipEventElapsedTime.argtypes=[C.POINTER(C.c_float),C.c_void_p,C.c_void_p]
ev=[C.c_void_p() for _ in range(10)]
for e in ev: hip.hipEventCreate(C.byref(e))
arr=(C.c_void_p*10)(*ev)
for it in range(300): gsr.forward_ws(s,ws,ins,None)   # clocks up
torch.cuda.synchronize()
tot=[]
for it in range(40):
    gsr.forward_ws(s,ws,ins,None,events=arr); torch.cuda.synchronize()
    ms=C.c_float(0); hip.hipEventElapsedTime(C.byref(ms),ev[0],ev[1]); tot.append(ms.value*1e3)
print(os.environ.get('GSR_LIB_OVERRIDE','default'), 'preprocess us: median %.1f min %.1f'%(np.median(tot), min(tot)))
