#!/usr/bin/env python3
"""CPU simulation of the rounding error of Winograd F(2x2, 3x3) in fp32 against the direct fp32 accumulation chain, both
judged by torch's fp64 convolution (max-norm relative, the metric of tests/test_c3_parity_gpu.py): the parity gate of the
round-5 experiment, run BEFORE any kernel existed.  fp32 transforms, sequential fp32 accumulation over
the input channels (output) / over the tiles (weight gradient); numpy, minutes on one core.

    python tools/wino_numerics_sim.py            # output / data gradient form
    python tools/wino_numerics_sim.py wgrad      # weight gradient form
Recorded (round 5): output  direct 1.1e-6 ... 3.0e-6, Winograd 4.3e-7 ... 1.6e-6;  weight gradient  direct 1.6e-6 ... 3.8e-6,
Winograd 6.9e-7 ... 2.0e-6."""
import sys
import numpy as np
import torch
import torch.nn.functional as F

torch.manual_seed(0)

def sim(B,K,N,S):
    g = torch.Generator().manual_seed(B*7+K*13+N*3+S+3+1)
    x = torch.randn(B,K,S,S,generator=g)
    w = torch.randn(N,K,3,3,generator=g)/(K*9)**0.5
    ref = F.conv2d(x.double(), w.double(), padding=1)
    # direct fp32 sequential chain over (k, tap)
    xp = F.pad(x,(1,1,1,1)).numpy()
    wn = w.numpy()
    acc = np.zeros((B,N,S,S),np.float32)
    for k in range(K):
        for t in range(9):
            dy,dx=t//3,t%3
            acc += wn[None,:,k,dy,dx,None,None]*xp[:,k,None,dy:dy+S,dx:dx+S]
    ed = np.abs(acc-ref.numpy()).max()/np.abs(ref.numpy()).max()
    # winograd F(2x2,3x3)
    G = np.array([[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]],np.float32)
    BT = np.array([[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]],np.float32)
    AT = np.array([[1,1,1,0],[0,1,-1,-1]],np.float32)
    U = np.einsum('ia,nkab,jb->ijnk',G.astype(np.float64),wn.astype(np.float64),G.astype(np.float64)).astype(np.float32)  # weight transform in higher precision then rounded
    U32 = np.einsum('ia,nkab->nkib',G,wn).astype(np.float32); U32=np.einsum('nkib,jb->ijnk',U32,G).astype(np.float32)
    T=S//2
    # tiles: d[b,k,ty,tx,4,4]
    d = np.lib.stride_tricks.sliding_window_view(xp,(4,4),axis=(2,3))[:,:,::2,::2]  # B,K,T,T,4,4
    t1 = np.einsum('ia,bkyxac->bkyxic',BT,d).astype(np.float32)
    V = np.einsum('bkyxic,jc->ijbkyx',t1,BT).astype(np.float32)
    out={}
    for name,UU in (('U64',U),('U32',U32)):
        M = np.zeros((4,4,B,N,T,T),np.float32)
        for k in range(K):
            M += UU[:,:,None,:,k,None,None]*V[:,:,:,k,None]
        y1 = np.einsum('pi,ijbnyx->pjbnyx',AT,M).astype(np.float32)
        Y = np.einsum('pjbnyx,qj->bnypxq',y1,AT).astype(np.float32).reshape(B,N,S,S)
        out[name]=np.abs(Y-ref.numpy()).max()/np.abs(ref.numpy()).max()
    print(f'B{B} K{K} N{N} S{S}: direct {ed:.2e} wino(U64) {out["U64"]:.2e} wino(U32) {out["U32"]:.2e}',flush=True)


def simw(B,K,N,S,nsplit=1):
    g = torch.Generator().manual_seed(B*7+K*13+N*3+S)
    x = torch.randn(B,K,S,S,generator=g); go = torch.randn(B,N,S,S,generator=g)
    ref = torch.nn.grad.conv2d_weight(x.double(), (N,K,3,3), go.double(), padding=1).numpy()
    xp = F.pad(x,(1,1,1,1)).numpy(); gon = go.numpy()
    # direct: sequential fp32 over pixels (vectorised over n,k,taps): order b,y,x
    acc = np.zeros((N,K,3,3),np.float32)
    for b in range(B):
        for y in range(S):
            # accumulate row by row sequentially over x
            for xx in range(S):
                patch = xp[b,:,y:y+3,xx:xx+3]          # K,3,3
                acc += gon[b,:,y,xx][:,None,None,None]*patch[None]
    ed = np.abs(acc-ref).max()/np.abs(ref).max()
    BT = np.array([[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]],np.float32)
    A = np.array([[1,0],[1,1],[1,-1],[0,-1]],np.float32)
    G = np.array([[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]],np.float32)
    T=S//2
    d = np.lib.stride_tricks.sliding_window_view(xp,(4,4),axis=(2,3))[:,:,::2,::2]
    V = np.einsum('ia,bkyxac,jc->bkyxij',BT,d,BT).astype(np.float32)
    gy = gon.reshape(B,N,T,2,T,2).transpose(0,1,2,4,3,5)  # B,N,T,T,2,2
    dM = np.einsum('ia,bnyxac,jc->bnyxij',A,gy,A).astype(np.float32)
    dU = np.zeros((N,K,4,4),np.float32)
    for b in range(B):
        for y in range(T):
            for xx in range(T):
                dU += dM[b,:,y,xx][:,None]*V[b,:,y,xx][None]
    dW = np.einsum('ia,nkij,jb->nkab',G,dU,G).astype(np.float32)
    ew = np.abs(dW-ref).max()/np.abs(ref).max()
    print(f'B{B} K{K} N{N} S{S}: direct {ed:.2e} wino {ew:.2e}',flush=True)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'wgrad':
        for cfg in [(2, 8, 8, 32), (4, 4, 4, 64), (8, 4, 4, 32)]:
            simw(*cfg)
    else:
        for cfg in [(2, 256, 16, 32), (2, 1024, 8, 16), (1, 128, 16, 64), (2, 512, 8, 32), (2, 2048, 4, 8)]:
            sim(*cfg)
