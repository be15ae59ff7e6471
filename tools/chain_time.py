import ctypes, os, sys, torch
REPO='/root/repo'
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO,'physics-aware-multiplex-gnn_amd'))
from pamnet_amd import lib
from pamnet_amd.fused import _parr, _iarr
D=128; dev=torch.device('cuda:0')
n=int(sys.argv[1]) if len(sys.argv)>1 else 2286
NW=15
W=[torch.randn(D,D,device=dev)*0.05 for _ in range(NW)]
b=[torch.zeros(D,device=dev) for _ in range(11)]
images=torch.empty(NW,D*D,device=dev)
lib.call('pamnet_pack_weights_f32', NW, _parr(W), _iarr([D]*NW), 0, lib.ptr(images), lib.stream_of(images))
img=[images[i] for i in range(NW)]
x2,rx=torch.randn(n,D,device=dev),torch.randn(n,D,device=dev)
w_out,b_out,w_att=torch.randn(D,device=dev),torch.zeros(1,device=dev),torch.randn(D,device=dev)
Z,R,xo=torch.empty(10,n,D,device=dev),torch.empty(2,n,D,device=dev),torch.empty(n,D,device=dev)
zx1,x1,P=torch.empty(n,D,device=dev),torch.empty(n,D,device=dev),torch.empty(4,n,D,device=dev)
for nblk in (0,2,4):
    def run():
        lib.call('pamnet_node_tail_fwd_f32', lib.ptr(x2), lib.ptr(rx), n, _parr(img[:10]), _parr(b[:10]), lib.ptr(w_out), lib.ptr(b_out), lib.ptr(w_att), lib.ptr(Z), lib.ptr(R), lib.ptr(xo), None, None,
                 lib.ptr(img[10]) if nblk else None, lib.ptr(b[10]) if nblk else None, _parr(img[11:11+nblk]) if nblk else None, D, nblk, lib.ptr(zx1) if nblk else None, lib.ptr(x1) if nblk else None, lib.ptr(P) if nblk else None, 1, lib.stream_of(x2))
    for _ in range(10): run()
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(200): run()
    e.record(); e.synchronize()
    print('n=%d fwd chain, packed, heads deferred, next head nblk=%d: %.2f us per launch (back to back)' % (n, nblk, s.elapsed_time(e)/200*1e3))
