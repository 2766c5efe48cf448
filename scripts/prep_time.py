import sys, os, time, numpy as np, torch, ctypes as C
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,R+'/tests')
from conftest import load_package
gsr=load_package(); syn=gsr.synthetic
cam=syn.make_camera(**syn.REPLICA); sc=syn.make_scene(1000000,cam,seed=0)
s=gsr.capi.Settings.from_camera(cam)
t=lambda x: torch.as_tensor(x,dtype=torch.float32,device='cuda').contiguous()
ins=dict(means3D=t(sc.means3D),opacities=t(sc.opacities),colors=t(sc.colors),shs=None,scales=t(sc.scales),rotations=t(sc.rotations),cov3D=None)
ws=gsr.capi.Workspace(1000000,1200,680,max_rendered=4000000)
hip=C.CDLL("libamdhip64.so.7"); hip.hipEventCreate.argtypes=[C.POINTER(C.c_void_p)]; hip.hipEventElapsedTime.argtypes=[C.POINTER(C.c_float),C.c_void_p,C.c_void_p]; hip.hipEventSynchronize.argtypes=[C.c_void_p]
ev=[C.c_void_p() for _ in range(10)]
for e in ev: hip.hipEventCreate(C.byref(e))
arr=(C.c_void_p*10)(*ev)
tot=np.zeros(5)
for it in range(25):
    gsr.forward_ws(s,ws,ins,None,events=arr); torch.cuda.synchronize()
    if it>=5:
        for k in range(5):
            ms=C.c_float(0); hip.hipEventElapsedTime(C.byref(ms),ev[2*k],ev[2*k+1]); tot[k]+=ms.value
print(os.environ.get('GSR_LIB_OVERRIDE','default'), 'us: preprocess %.1f scan %.1f fill %.1f sort %.1f blend %.1f'%tuple(tot/20*1e3))
