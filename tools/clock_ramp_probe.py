"""What slows the stage-I weight-gradient GEMM (TN, K = B*L) inside a small-shard step against the same launch back to back:
the launch behind nothing / 600 tiny launches / the K = B part-A weight-gradient pair / 20 ms of idle, and with dense, scaled
and sparse dP operands (power-dependent clocks).  profiles/r04_small_shards.md.   python tools/clock_ramp_probe.py [B]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from recurrent_fusion_network_amd import _native as nv
dev='cuda'; L,D,A,T=196,2048,512,8
B=int(sys.argv[1]) if len(sys.argv)>1 else 32
BL=B*L
ws=torch.empty(256<<20,dtype=torch.uint8,device=dev)
X=torch.randn(BL,D,device=dev); P=torch.randn(T,BL,A,device=dev); dW=[torch.empty(A,D,device=dev) for _ in range(T)]
tn=[(dW[t],D,[(P[t],A,0,X,D,0,BL,None)]) for t in range(T)]
G=torch.randn(T,B,2048,device=dev); H=torch.randn(T,B,2048,device=dev); dWa=[torch.empty(2048,2048,device=dev) for _ in range(T)]
pa=[(dWa[t],2048,[(G[t],2048,0,H[t],2048,0,B,None)]) for t in range(T)]
small=torch.zeros(64,device=dev)
def run_tn(): nv.gemm(A,D,tn,ws=ws)
def t(pre,reps=10):
    tot=0
    for i in range(reps+2):
        pre(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record(); run_tn(); e1.record(); torch.cuda.synchronize()
        if i>=2: tot+=e0.elapsed_time(e1)
    return tot/reps*1e3
def tiny():
    for _ in range(600): small.add_(1.0)
def parta():
    nv.gemm(2048,2048,pa,ws=ws); nv.gemm(2048,2048,pa,ws=ws)
def idle():
    torch.cuda.synchronize(); import time; time.sleep(0.02)
print('B=%d TN: nothing before %.1f us | 600 tiny launches before %.1f | part-A pair before %.1f | 20 ms idle before %.1f' % (B, t(lambda:None), t(tiny), t(parta), t(idle)))
P.mul_(1e-6)
print('  with 1e-6-scaled dP: nothing before %.1f us | part-A pair before %.1f' % (t(lambda:None), t(parta)))
P.zero_(); P[:, ::7, ::5]=1e-4
print('  with sparse dP: nothing before %.1f us' % t(lambda:None))
# controlled A/B: dense and sparse operands alternate, every launch behind the same K = B weight-gradient pair
Pd = torch.randn(T, BL, A, device=dev)
Ps = torch.zeros(T, BL, A, device=dev); Ps[:, ::7, ::5] = 1e-4
Xs = torch.zeros(BL, D, device=dev); Xs[::3, ::4] = 1.0
def tn_of(Pm, Xm): return [(dW[t_], D, [(Pm[t_], A, 0, Xm, D, 0, BL, None)]) for t_ in range(T)]
cases = [('dense dP, dense X', tn_of(Pd, X)), ('sparse dP, dense X', tn_of(Ps, X)), ('sparse dP, sparse X', tn_of(Ps, Xs))]
acc = {n: [] for n, _ in cases}
for rnd in range(6):
    for n, pr in cases:
        parta(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); nv.gemm(A, D, pr, ws=ws); e1.record(); torch.cuda.synchronize()
        if rnd: acc[n].append(e0.elapsed_time(e1) * 1e3)
print('  alternating, each behind the part-A pair: ' + ' | '.join('%s %.1f us (min %.1f)' % (n, sum(v) / len(v), min(v)) for n, v in acc.items()))
