import sys, os, time, numpy as np, torch
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,R+'/tests')
from conftest import load_package
gsr=load_package(); syn=gsr.synthetic
for P,mult,mode in ((300000,4.0,'rgb'),(1000000,4.0,'depth'),(2000000,1.0,'rgb')):
    cam=syn.make_camera(**syn.REPLICA); sc=syn.make_scene(P,cam,seed=0,scale_mult=mult,color_mode=mode)
    s=gsr.capi.Settings.from_camera(cam)
    t=lambda x: torch.as_tensor(x,dtype=torch.float32,device='cuda').contiguous()
    ins=dict(means3D=t(sc.means3D),opacities=t(sc.opacities),colors=t(sc.colors),shs=None,scales=t(sc.scales),rotations=t(sc.rotations),cov3D=None)
    g=t(sc.dL_dpix)
    st0=gsr.forward(s,ins['means3D'],ins['opacities'],colors=ins['colors'],scales=ins['scales'],rotations=ins['rotations']); Rr=st0.num_rendered
    d=gsr.debug_export(st0); cnt=(d['ranges'][:,1]-d['ranges'][:,0]); del st0
    ws=gsr.capi.Workspace(P,cam.width,cam.height,max_rendered=int(Rr*1.1)+1024)
    gr=gsr.capi.alloc_grads(P,0,'cuda')
    def step():
        st=gsr.forward_ws(s,ws,ins,None); gsr.backward(st,g,grads=gr)
    for _ in range(3): step()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/10
    print(f"P={P} x{mult} {mode}: R={Rr} (R/P={Rr/P:.1f}, max tile {cnt.max()}, mean {cnt.mean():.0f}) {dt*1e3:.2f} ms/step")
