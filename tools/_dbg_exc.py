import sys; sys.path.insert(0,'.')
import numpy as np, torch
from oracle import pb_oracle as O
from pb_llm_amd import synth
from pb_llm_amd import quant as Q
from pb_llm_amd.packing import pack_dense
N,K=32,1024
rng=np.random.default_rng(1)
W=np.where(rng.random((N,K))<0.5,0.25,-0.125).astype(np.float32)
hi=np.full((N,1),0.25,np.float32); lo=np.full((N,1),-0.125,np.float32)
ss=np.full(N,0.01,np.float32); sz=np.full(N,100.0,np.float32)
x=synth.activations((1,K),3,21)
def run(W,tag):
    p=pack_dense(W,hi,lo,ss,sz)
    y=Q.PBLinear(p.to('cuda'),None)(torch.from_numpy(x).cuda()).float().cpu().numpy()
    ref=O.dense_linear(x,W)
    err=np.abs(y-ref)[0]
    print(tag,"nnz",p.nnz,"nexc",p.nexc,"maxerr",err.max(),"rows bad",np.nonzero(err>1e-2)[0][:20])
run(W,"plain")
W1=W.copy(); W1[5,100]=1.2345; run(W1,"one exc row5")
W2=W.copy(); W2[5,100]=ss[5]*(50-100); run(W2,"one code row5")
W3=W.copy(); 
for r in range(N): W3[r,7*r+3]=0.777+r
run(W3,"exc per row")
W4=W.copy(); W4[20,:]=ss[20]*(rng.integers(0,256,K).astype(np.float32)-100); run(W4,"dense row20")
W5=W.copy(); W5[0,[3,900,901]]=ss[0]*(np.array([7,250,0],np.float32)-100); run(W5,"gaps")
