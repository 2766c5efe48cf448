"""Host-side cost of the libtorch entry points alone (no autograd, no loss): calls per second of
_C.rasterize_gaussians + _C.rasterize_gaussians_backward next to the GPU time of the same work."""
import sys, os, time, torch
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,R+'/tests'); sys.path.insert(0,R+'/gsorb-slam_amd')
from conftest import load_package
gsr=load_package(); syn=gsr.synthetic
import diff_gaussian_rasterization as dgr
C=dgr._C
for P in (30000, 300000):
    cam=syn.make_camera(**syn.REPLICA); sc=syn.make_scene(P,cam,seed=0)
    t=lambda a: torch.tensor(a,dtype=torch.float32,device='cuda')
    e=torch.empty(0,device='cuda')
    bg,m,col,op,scl,rot,vm,pm,cp=t(cam.bg),t(sc.means3D),t(sc.colors),t(sc.opacities),t(sc.scales),t(sc.rotations),t(cam.viewmatrix),t(cam.projmatrix),t(cam.campos)
    g=t(sc.dL_dpix)
    def step():
        nr,color,radii,geom,binn,img,depth=C.rasterize_gaussians(bg,m,col,op,scl,rot,1.0,e,vm,pm,cam.tanfovx,cam.tanfovy,cam.height,cam.width,e,0,cp,False)
        return C.rasterize_gaussians_backward(bg,m,radii,col,scl,rot,1.0,e,vm,pm,cam.tanfovx,cam.tanfovy,g,e,0,cp,geom,nr,binn,img)
    for _ in range(10): step()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(200): step()
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/200
    print(f"_C fwd+bwd P={P}: {dt*1e6:.1f} us/step")
