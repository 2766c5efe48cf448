"""The sync-free path (gsr_forward_ws + gsr_backward: no allocation, no host read) can be captured in a HIP graph and
replayed: the replay reproduces the eager results. (Replay is not faster — scripts/graph_replay.py: 55.7 vs 59.5 us at
10 k splats, 509 vs 508 us at 1 M — the step is bound by the GPU-side dispatch of its nine dependent kernels, not by
the host; what capture buys a caller is one launch per step inside a larger captured loop.)"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_forward_ws_and_backward_replay_from_a_hip_graph(gsr, syn):
    cam = syn.make_camera(**syn.TUM1)
    sc = syn.make_scene(20000, cam, seed=3)
    s = gsr.capi.Settings.from_camera(cam, device="cuda")
    t = lambda x: torch.as_tensor(x, dtype=torch.float32, device="cuda").contiguous()
    ins = dict(means3D=t(sc.means3D), opacities=t(sc.opacities), colors=t(sc.colors), shs=None, scales=t(sc.scales),
               rotations=t(sc.rotations), cov3D=None)
    g_in = t(sc.dL_dpix)
    st0 = gsr.forward(s, ins["means3D"], ins["opacities"], colors=ins["colors"], scales=ins["scales"], rotations=ins["rotations"])
    ws = gsr.capi.Workspace(20000, cam.width, cam.height, max_rendered=int(st0.num_rendered * 1.25) + 1024, device="cuda")
    grads = gsr.capi.alloc_grads(20000, 0, "cuda", intermediates=False)

    def step():
        st = gsr.forward_ws(s, ws, ins, None)
        gsr.backward(st, g_in, grads=grads, once=True)
        return st

    st = step()
    torch.cuda.synchronize()
    col, dmean, dop = st.color.clone(), grads.dL_dmeans3D.clone(), grads.dL_dopacity.clone()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        st = step()
    # new inputs in the same buffers: the replay must render THEM
    ins["means3D"].add_(torch.tensor([0.01, -0.02, 0.03], device="cuda"))
    st.color.zero_(); grads.dL_dmeans3D.zero_()
    graph.replay()
    torch.cuda.synchronize()
    col_g, dmean_g, dop_g = st.color.clone(), grads.dL_dmeans3D.clone(), grads.dL_dopacity.clone()
    st2 = step()
    torch.cuda.synchronize()
    assert not torch.equal(col_g, col)                                      # the moved scene, not the captured one
    assert torch.equal(col_g, st2.color)                                    # forward: bit-identical to the eager call
    scale = float(grads.dL_dmeans3D.abs().max())
    assert float((dmean_g - grads.dL_dmeans3D).abs().max()) <= 1e-5 * scale  # backward: float atomics, order not fixed
    assert float((dop_g - grads.dL_dopacity).abs().max()) <= 1e-5 * float(grads.dL_dopacity.abs().max())
    n, ovf = ws.status()
    assert not ovf and n > 0
