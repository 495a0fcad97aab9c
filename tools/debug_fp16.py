import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import svdq_oracle as O
from tests.helpers import make_module, t16, f32
M,K,N,R=512,3072,384,32
dtype='fp16'
L=O.make_svdq_layer(K,N,R,seed=K+N,dtype=dtype,bias=True,cheap=True)
x=O.make_activations(M,K,seed=K+N,dtype=dtype)
mod=make_module(L,dtype)
y=f32(mod(t16(x,dtype).view(1,M,K)))[0]
ref=O.svdq_linear(x,L,dtype,'fp32')['out']
err=np.abs(y-ref)
idx=np.argwhere(err>2**-10*np.abs(ref)+1e-4)
print('n bad',len(idx))
q,a,la=O.quantize_w4a4_act_fuse_lora(x,L['smooth'],L['proj_down'],dtype)
qx,asc,lag=mod.quantize(t16(x,dtype))
print('lora_act max abs', np.abs(la).max(), 'gpu-vs-oracle max diff', np.abs(lag.cpu().numpy()-la).max())
la16=O.round16(la,dtype); lag16=O.round16(lag.cpu().numpy(),dtype)
print('la16 mismatches', (la16!=lag16).sum(), 'max diff', np.abs(la16-lag16).max(), 'inf?', np.isinf(la16).sum())
for (i,j) in idx[:10]:
    print(i,j,'got',y[i,j],'ref',ref[i,j],'err',err[i,j], 'la row max', np.abs(la[i]).max())
