"""Is the sync-free path (gsr_forward_ws + gsr_backward) capturable in a HIP graph, and what does replaying it buy?
Captures one fwd+bwd step with torch.cuda.CUDAGraph (hipStreamBeginCapture underneath) and compares eager / replay."""
import sys, os, time, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R + '/tests')
from conftest import load_package
gsr = load_package(); syn = gsr.synthetic
for P, camname in ((10_000, "tum"), (100_000, "tum"), (300_000, "replica"), (1_000_000, "replica")):
    cam = syn.make_camera(**syn.CAMERAS[camname]); sc = syn.make_scene(P, cam, seed=0)
    s = gsr.capi.Settings.from_camera(cam, device="cuda")
    t = lambda x: torch.as_tensor(x, dtype=torch.float32, device="cuda").contiguous()
    ins = dict(means3D=t(sc.means3D), opacities=t(sc.opacities), colors=t(sc.colors), shs=None, scales=t(sc.scales), rotations=t(sc.rotations), cov3D=None)
    g_in = t(sc.dL_dpix)
    st0 = gsr.forward(s, ins["means3D"], ins["opacities"], colors=ins["colors"], scales=ins["scales"], rotations=ins["rotations"])
    Rn = st0.num_rendered; del st0
    ws = gsr.capi.Workspace(P, cam.width, cam.height, max_rendered=int(Rn * 1.25) + 1024, device="cuda")
    grads = gsr.capi.alloc_grads(P, 0, "cuda", intermediates=False)
    def step():
        st = gsr.forward_ws(s, ws, ins, None)
        gsr.backward(st, g_in, grads=grads, once=True)
        return st
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): step()
    torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 200
    ref = grads.dL_dmeans3D.clone(); st = step(); torch.cuda.synchronize(); col = st.color.clone()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(3): step()
    torch.cuda.current_stream().wait_stream(side)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        st = step()
    for _ in range(5): gr.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): gr.replay()
    torch.cuda.synchronize(); rep = (time.perf_counter() - t0) / 200
    same = bool(torch.equal(st.color, col)) and float((grads.dL_dmeans3D - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
    print(f"P={P} {camname}: eager {eager*1e6:.1f} us/step, graph replay {rep*1e6:.1f} us/step, results equal: {same}")
