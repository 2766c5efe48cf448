"""The SSIM kernels alone at 1200x680x3 (the plain pair gsr_ssim_forward / gsr_ssim_backward through capi.ssim_mean's pieces): us per launch
between torch events over 200 launches each, clocks up first. GSR_LIB_OVERRIDE picks the library."""
import sys, os, ctypes as C, numpy as np, torch
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,R+'/tests')
from conftest import load_package
gsr=load_package(); capi=gsr.capi; L=capi.lib()
H,W,Cc=680,1200,3
torch.manual_seed(0)
a=torch.rand(Cc,H,W,device='cuda'); b=(a+0.1*torch.randn_like(a)).clamp(0,1)
taps=gsr.harness._ssim_taps() if hasattr(gsr,'harness') and hasattr(gsr.harness,'_ssim_taps') else None
if taps is None:
    x=np.arange(11)-5; g=np.exp(-x**2/(2*1.5**2)); taps=(g/g.sum()).tolist()
tp=(C.c_float*11)(*[float(x) for x in taps])
partial=torch.empty((int(L.gsr_ssim_partials(Cc,H,W)),),device='cuda'); dmaps=torch.empty(3,Cc,H,W,device='cuda'); out=torch.empty_like(a); g1=torch.ones(1,device='cuda')
p=capi._p; st=capi._stream
def fwd(): capi._check(L.gsr_ssim_forward(p(a),p(b),Cc,H,W,tp,p(partial),p(dmaps),st()))
def bwd(): capi._check(L.gsr_ssim_backward(p(a),p(b),p(dmaps),Cc,H,W,tp,p(g1),p(out),st()))
# the mapping loss's variants (the pixel terms riding on the SSIM passes): gsr_map_loss_forward / _backward
dep=torch.rand(H,W,device='cuda')*3+0.5; sur=dep.clone(); sil=torch.rand(H,W,device='cuda'); fd=dep+0.05*torch.randn_like(dep); fd[::7]=0
np6=int(L.gsr_ssim_partials(3,H,W)); partial6=torch.empty((np6*6,),device='cuda'); dm=torch.empty(3,3,H,W,device='cuda')
w3=(C.c_float*3)(0.8,0.7,0.35); sums=torch.ones(8,device='cuda')*1000; neg=torch.tensor([-0.2],device='cuda'); gi=torch.empty_like(a); gd=torch.empty_like(dep)
def mfwd(): capi._check(L.gsr_map_loss_forward(p(a),p(dep),p(sur),p(sil),p(b),p(fd),H,W,tp,0.99,p(partial6),p(dm),st()))
def mbwd(): capi._check(L.gsr_map_loss_backward(p(a),p(dep),p(b),p(fd),p(dm),H,W,tp,w3,p(neg),p(sums),p(gi),p(gd),st()))
for _ in range(300): fwd(); bwd(); mfwd(); mbwd()
torch.cuda.synchronize()
res={}
for name,fn in (("fwd",fwd),("bwd",bwd),("mfwd",mfwd),("mbwd",mbwd)):
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): fn()
    e1.record(); torch.cuda.synchronize(); res[name]=e0.elapsed_time(e1)/200*1e3
e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): fwd(); bwd(); mfwd(); mbwd()
e1.record(); torch.cuda.synchronize()
print("interleaved (four different kernels in turn: cold instruction caches): %.1f us per round of four, sum of the four alone %.1f"%(e0.elapsed_time(e1)/200*1e3, sum(res.values())))
print(os.environ.get('GSR_LIB_OVERRIDE','default'), "ssim fwd %.1f us  bwd %.1f us   map-loss fwd %.1f us  bwd %.1f us   (sum %.6f)"%(res['fwd'],res['bwd'],res['mfwd'],res['mbwd'],float(partial.sum())/(Cc*H*W)))
# does it matter that the inputs were just written by another kernel (as in the loop: the render is the blend kernel's output)?
for label, pre in (("inputs rewritten before every launch", lambda: (a.mul_(1.0), dep.mul_(1.0), sur.mul_(1.0))), ("inputs untouched", lambda: None)):
    tot = 0.0
    for _ in range(100):
        pre(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record(); mfwd(); e1.record(); torch.cuda.synchronize(); tot += e0.elapsed_time(e1)
    print("map-loss fwd, %s: %.1f us (events around single launches)" % (label, tot / 100 * 1e3))
for name, fn in (("plain fwd", fwd), ("plain bwd", bwd), ("map-loss bwd", mbwd)):
    for label, pre in (("rewritten", lambda: (a.mul_(1.0), dm.mul_(1.0), dmaps.mul_(1.0))), ("untouched", lambda: None)):
        tot = 0.0
        for _ in range(100):
            pre(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize(); tot += e0.elapsed_time(e1)
        print("%s, inputs %s: %.1f us" % (name, label, tot / 100 * 1e3))
