"""dmaps / gradient of the SSIM kernels of two libraries on the same images, row by row (GSR_LIB_A / GSR_LIB_B: paths; run on the GPU box)."""
import sys, os, ctypes as C, numpy as np, torch, subprocess, json
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "one":
    sys.path.insert(0,R); sys.path.insert(0,R+'/tests')
    from conftest import load_package
    gsr=load_package(); capi=gsr.capi; L=capi.lib()
    H,W,Cc=int(sys.argv[2]),int(sys.argv[3]),3
    torch.manual_seed(0)
    a=torch.rand(Cc,H,W,device='cuda'); b=(a+0.1*torch.randn_like(a)).clamp(0,1)
    x=np.arange(11)-5; g=np.exp(-x**2/(2*1.5**2)); taps=(g/g.sum()).tolist()
    tp=(C.c_float*11)(*[float(x) for x in taps])
    partial=torch.zeros((int(L.gsr_ssim_partials(Cc,H,W)),),device='cuda'); dmaps=torch.zeros(3,Cc,H,W,device='cuda'); out=torch.zeros_like(a); g1=torch.ones(1,device='cuda')
    p=capi._p; st=capi._stream
    capi._check(L.gsr_ssim_forward(p(a),p(b),Cc,H,W,tp,p(partial),p(dmaps),st()))
    capi._check(L.gsr_ssim_backward(p(a),p(b),p(dmaps),Cc,H,W,tp,p(g1),p(out),st()))
    torch.cuda.synchronize()
    torch.save({"dmaps":dmaps.cpu(),"grad":out.cpu(),"sum":float(partial.sum())}, sys.argv[4])
    sys.exit(0)
H,W=(int(sys.argv[1]),int(sys.argv[2])) if len(sys.argv)>2 else (70,130)
res={}
for tag in ("A","B"):
    env=dict(os.environ); env["GSR_LIB_OVERRIDE"]=os.environ["GSR_LIB_"+tag]
    subprocess.run([sys.executable,__file__,"one",str(H),str(W),f"/tmp/ssim_{tag}.pt"],env=env,check=True)
    res[tag]=torch.load(f"/tmp/ssim_{tag}.pt")
print("sum",res["A"]["sum"],res["B"]["sum"])
for k in ("dmaps","grad"):
    d=(res["A"][k]-res["B"][k]).abs()
    print(k,"max diff",float(d.max()),"scale",float(res["A"][k].abs().max()))
    rows=d.reshape(-1,H,W).amax(dim=(0,2)); cols=d.reshape(-1,H,W).amax(dim=(0,1))
    print(" rows with diff > 1e-6*scale:",[int(i) for i in torch.nonzero(rows>1e-5*float(res["A"][k].abs().max())).flatten()][:40])
    print(" cols:",[int(i) for i in torch.nonzero(cols>1e-5*float(res["A"][k].abs().max())).flatten()][:40])
