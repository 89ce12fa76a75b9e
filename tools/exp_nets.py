"""Experiment: forward time of the Value net at B=1024 in several PyTorch forms."""
import time, sys, os
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iago_amd import network

def bench(fn, x, n=20):
    for _ in range(3): fn(x)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): fn(x)
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3

B=int(sys.argv[1]) if len(sys.argv)>1 else 1024
torch.manual_seed(0)
v = network.Value().cuda().eval()
x = (torch.rand(B,2,8,8,device='cuda')>0.5).float()
FL = 122.99e6*B
network.Block.fused_inference=False
with torch.no_grad():
    t=bench(v,x); print('default NCHW fp32: %.3f ms  %.1f TF'%(t,FL/t/1e9))
    network.Block.fused_inference=True
    t=bench(v,x); print('fused conv+bias+relu: %.3f ms  %.1f TF'%(t,FL/t/1e9))
    network.Block.fused_inference=False
    a0=v(x); network.Block.fused_inference=True; a1=v(x); print('fused vs unfused max diff', (a0-a1).abs().max().item()); network.Block.fused_inference=False
    torch.backends.cudnn.benchmark=True
    t=bench(v,x); print('benchmark=True: %.3f ms  %.1f TF'%(t,FL/t/1e9))
    v2 = network.Value().cuda().eval().to(memory_format=torch.channels_last)
    xc = x.contiguous(memory_format=torch.channels_last)
    t=bench(v2,xc); print('channels_last: %.3f ms  %.1f TF'%(t,FL/t/1e9))
    # unfold + matmul
    Ws=[getattr(v,'block%d'%k).conv.weight for k in range(1,10)]
    bs=[getattr(v,'block%d'%k).conv.bias for k in range(1,10)]
    Wm=[w.reshape(w.shape[0],-1).t().contiguous() for w in Ws]
    def unf(x):
        h=x
        for k in range(9):
            cols=F.unfold(h,3,padding=1)            # (B, C*9, 64)
            h=torch.relu(cols.transpose(1,2).reshape(-1,cols.shape[1]) @ Wm[k] + bs[k])  # (B*64, O)
            h=h.reshape(x.shape[0],64,-1).transpose(1,2).reshape(x.shape[0],-1,8,8)
        return h
    t=bench(unf,x); print('unfold+matmul: %.3f ms  %.1f TF'%(t,FL/t/1e9))
    ref=v.block9(v.block8(v.block7(v.block6(v.block5(v.block4(v.block3(v.block2(v.block1(x)))))))))
    print('unfold err', (unf(x)-ref).abs().max().item())
    # bf16 autocast for reference
    with torch.autocast('cuda',dtype=torch.bfloat16):
        t=bench(v,x); print('bf16 autocast: %.3f ms  %.1f TF'%(t,FL/t/1e9))
    # plain big GEMM to see fp32 hipBLASLt rate
    a=torch.randn(65536,1152,device='cuda'); bm=torch.randn(1152,128,device='cuda')
    t=bench(lambda q: a@bm, None); print('GEMM 65536x1152x128 fp32: %.3f ms %.1f TF'%(t, 2*65536*1152*128/t/1e9))
