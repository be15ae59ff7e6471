import ctypes, os, sys
REPO='/root/repo'
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO,'physics-aware-multiplex-gnn_amd'))
import torch
from pamnet_amd import lib
dev=torch.device('cuda:0'); D=128
def t_us(fn, reps=100):
    for _ in range(5): fn()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)*1e3/reps
for rows,K,two,dx in ((32888,16,False,True),(4316,16,False,True),(17640,42,True,False),(17640,42,False,False),(32888,16,False,False),(65776,16,False,True),(16444,16,False,True)):
    x=torch.randn(rows,K,device=dev); g=torch.randn(rows,D,device=dev)
    W0,b0=torch.randn(D,K,device=dev),torch.randn(D,device=dev)
    W1,b1=(torch.randn(D,K,device=dev),torch.randn(D,device=dev)) if two else (None,None)
    kind=(torch.arange(rows,device=dev)%3==0).to(torch.int32) if two else None
    need=ctypes.c_int64(0); lib.call('pamnet_embed_scratch_floats', rows, K, ctypes.addressof(need))
    partial=torch.empty(need.value,device=dev)
    dW0,db0=torch.empty_like(W0),torch.empty_like(b0)
    dW1,db1=(torch.empty_like(W0),torch.empty_like(b0)) if two else (None,None)
    dxo=torch.empty(rows,K,device=dev) if dx else None
    st=lib.stream_of(x)
    def run():
        lib.call('pamnet_embed_bwd_f32', lib.ptr(x), rows, K, lib.ptr(kind), lib.ptr(W0), lib.ptr(b0), lib.ptr(W1), lib.ptr(b1), 1, lib.ptr(g), lib.ptr(dW0), lib.ptr(db0), lib.ptr(dW1), lib.ptr(db1), lib.ptr(dxo), lib.ptr(partial), st)
    print('rows %6d K %2d two %d dx %d: %.1f us (bwd + reduce launches)' % (rows,K,two,dx,t_us(run)))
