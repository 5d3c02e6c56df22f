import sys; sys.path.insert(0,'.')
import numpy as np, torch
from oracle import pb_oracle as O
from pb_llm_amd import synth
from pb_llm_amd import quant as Q
from pb_llm_amd.packing import pack_dense
N,K=16,512
rng=np.random.default_rng(1)
hi=np.full((N,1),0.25,np.float32); lo=np.full((N,1),-0.125,np.float32)
ss=np.full(N,0.01,np.float32); sz=np.full(N,100.0,np.float32)
def run(W,x,tag):
    p=pack_dense(W,hi,lo,ss,sz)
    y=Q.PBLinear(p.to('cuda'),None)(torch.from_numpy(x).cuda()).float().cpu().numpy()
    ref=O.dense_linear(x,W)
    print(tag,"nnz",p.nnz,"y",np.round(y[0,:16],3),"\n   ref",np.round(ref[0,:16],3))
x1=np.ones((1,K),np.float16)
W=np.full((N,K),-0.125,np.float32); run(W,x1,"all lo, x=1")
W=np.full((N,K),0.25,np.float32); run(W,x1,"all hi, x=1")
W=np.full((N,K),-0.125,np.float32)
for r in range(N): W[r,:32*(r+1)]=0.25
run(W,x1,"ramp rows, x=1")
x=synth.activations((1,K),3,21)
W=np.where(rng.random((N,K))<0.5,0.25,-0.125).astype(np.float32); run(W,x,"random, no salient")
W2=W.copy(); W2[3,10]=ss[3]*(50-100); W2[3,11]=ss[3]*(150-100); run(W2,x,"2 codes row3")
W3=W.copy()
for r in range(N):
    for c in range(0,K,7): W3[r,c]=ss[r]*((c%200)-100)
run(W3,x,"many codes")
