import sys, torch, math
sys.path.insert(0, "/root/repo")
from cpc_audio_amd import _hip
dev="cuda:0"; B,V,H=256,100,256; bf=torch.bfloat16
g=torch.Generator().manual_seed(0)
w=torch.randn(3*H,H,generator=g)/math.sqrt(H); b=torch.randn(3*H,generator=g)*0.1
Gi=torch.randn(B,V,3*H,generator=g).to(dev); dc=torch.randn(B,H,generator=g).to(dev)
dW,db=w.to(dev),b.to(dev)
wf=torch.empty(3*H*H,device=dev,dtype=bf); wt=torch.empty(3*H*H,device=dev,dtype=bf)
_hip.call("cpc_prep_frag",_hip.ptr(dW),_hip.ptr(wf),3*H,H,H,0,1); _hip.call("cpc_prep_frag",_hip.ptr(dW),_hip.ptr(wt),H,3*H,H,1,1)
Gi=Gi.to(bf)
Hall=torch.empty(B,V+1,H,device=dev,dtype=bf); tape=torch.zeros(_hip.lib().cpc_gru_tape_elems(B,V,H,1),device=dev,dtype=bf); c=torch.empty(B,H,device=dev)
dGi=torch.empty(B,V,4*H,device=dev,dtype=bf)
def run(nw):
    _hip.lib().cpc_gru_set_streaming(nw)
    res=[]
    for name,fn in (("fwd",lambda:_hip.call("cpc_gru_fwd",_hip.ptr(Gi),_hip.ptr(wf),_hip.ptr(db),_hip.ptr(Hall),_hip.ptr(tape),_hip.ptr(c),B,V,H,1)),
                    ("bwd",lambda:_hip.call("cpc_gru_bwd",_hip.ptr(dc),_hip.ptr(tape),_hip.ptr(wt),_hip.ptr(dGi),B,V,H,1))):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        res.append((name, e0.elapsed_time(e1)/10))
    return res, c.clone(), dGi.clone()
r8,c8,g8=run(100)
print("resident", r8)
for dbg in (1, 2, 4, 5, 7):
    _hip.lib().cpc_gru_set_streaming(100 + dbg)
    print("bwd debug bits", dbg, run(100 + dbg)[0][1])
_hip.lib().cpc_gru_set_streaming(100)
