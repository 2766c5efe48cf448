"""End-to-end latency of the drop-in Python operator (torch allocations + the one D2H sync included)
next to the sync-free C-ABI loop bench.py times."""
import sys, os, time, numpy as np, torch
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,R+'/tests'); sys.path.insert(0,R+'/gsorb-slam_amd')
from conftest import load_package
gsr=load_package(); syn=gsr.synthetic
import diff_gaussian_rasterization as dgr
for P in (30000, 300000, 1000000):
    cam=syn.make_camera(**syn.REPLICA); sc=syn.make_scene(P,cam,seed=0)
    t=lambda a: torch.tensor(a,dtype=torch.float32,device='cuda')
    st=dgr.GaussianRasterizationSettings(cam.height,cam.width,cam.tanfovx,cam.tanfovy,t(cam.bg),1.0,t(cam.viewmatrix),t(cam.projmatrix),0,t(cam.campos),False)
    r=dgr.GaussianRasterizer(st)
    prm=[t(sc.means3D).requires_grad_(True), t(sc.opacities).requires_grad_(True), t(sc.colors).requires_grad_(True), t(sc.scales).requires_grad_(True), t(sc.rotations).requires_grad_(True)]
    g=t(sc.dL_dpix)
    def step():
        m2=torch.zeros_like(prm[0],requires_grad=True)
        im,rad,dep=r(means3D=prm[0],means2D=m2,opacities=prm[1],colors_precomp=prm[2],scales=prm[3],rotations=prm[4])
        (im*g).sum().backward()
        for p in prm: p.grad=None
    for _ in range(5): step()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(30): step()
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/30
    print(f"python op fwd+bwd P={P}: {dt*1e3:.3f} ms/step  -> {P*cam.width*cam.height/dt:.3e} splats*pixels/s")
