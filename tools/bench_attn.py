"""Times the stage-I attention kernels alone at the C3 shapes (B=256, L=196, A=512, D=2048, 4 encoders round-robin
so nothing is served from the Infinity Cache) and prints achieved HBM GB/s per kernel.  Diagnostic only.
    python tools/bench_attn.py [--reps 20]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recurrent_fusion_network_amd import _native as n  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--B', type=int, default=256)
    ap.add_argument('--L', type=int, default=196)
    ap.add_argument('--A', type=int, default=512)
    ap.add_argument('--D', type=int, default=2048)
    ap.add_argument('--T1', type=int, default=8)
    ap.add_argument('--contig', action='store_true', help='step-major projection slices (row stride A) instead of the path layout (row stride T1*A)')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    B, L, A, D, T1, M = a.B, a.L, a.A, a.D, a.T1, 4
    x = [torch.randn(B, L, D, device=dev) for _ in range(M)]
    proj = [torch.randn(B, L, T1 * A, device=dev) for _ in range(M)]   # step-t slice has row stride T1*A
    PSL = A if a.contig else T1 * A          # row stride of one step's slice
    PSB = L * PSL
    hp = torch.randn(B, A, device=dev)
    w = torch.randn(A, device=dev) * 0.1
    bo = torch.zeros(1, device=dev)
    al = torch.softmax(torch.randn(B, L, device=dev), 1).contiguous()
    dal = torch.empty(B, L, device=dev)
    z = torch.empty(B, D, device=dev)
    dz = torch.randn(B, D, device=dev)
    dhp = torch.empty(B, A, device=dev)
    dwp = torch.empty(B, A, device=dev)
    st = n.stream_ptr()
    L_ = n.lib

    def scores(i):
        n.check(L_.rfn_attn_scores_fwd(proj[i].data_ptr(), PSB, PSL, hp.data_ptr(), w.data_ptr(),
                                       bo.data_ptr(), B, L, A, al.data_ptr(), st))

    def context(i):
        n.check(L_.rfn_attn_context_fwd(x[i].data_ptr(), L * D, D, al.data_ptr(), B, L, D, z.data_ptr(), D, st))

    def dalpha(i):
        n.check(L_.rfn_attn_context_bwd_dalpha(x[i].data_ptr(), L * D, D, dz.data_ptr(), D, B, L, D, dal.data_ptr(),
                                               st))

    def scores_bwd(i):
        n.check(L_.rfn_attn_scores_bwd(proj[i].data_ptr(), PSB, PSL, hp.data_ptr(), w.data_ptr(),
                                       al.data_ptr(), dal.data_ptr(), B, L, A, proj[i].data_ptr(), PSB, PSL,
                                       0, dhp.data_ptr(), dwp.data_ptr(), st))

    def fused_bwd(i):
        n.check(L_.rfn_attn_bwd(proj[i].data_ptr(), PSB, PSL, hp.data_ptr(), w.data_ptr(), al.data_ptr(),
                                x[i].data_ptr(), L * D, D, dz.data_ptr(), D, B, L, A, D, proj[i].data_ptr(),
                                PSB, PSL, 0, dhp.data_ptr(), dwp.data_ptr(), st))

    def split_bwd(i):
        dalpha(i)
        scores_bwd(i)

    cases = [('scores_fwd (raw+softmax)', scores, B * L * A * 4), ('context_fwd', context, B * L * D * 4),
             ('dalpha', dalpha, B * L * D * 4), ('scores_bwd (in place)', scores_bwd, 2 * B * L * A * 4),
             ('dalpha + scores_bwd (2 launches)', split_bwd, B * L * D * 4 + 2 * B * L * A * 4),
             ('rfn_attn_bwd (fused)', fused_bwd, B * L * D * 4 + 2 * B * L * A * 4)]
    for name, fn, nbytes in cases:
        for i in range(M):
            fn(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for r in range(a.reps):
            fn(r % M)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / a.reps
        print(f'{name:34s} {us:8.1f} us   {nbytes / us / 1e3:8.1f} GB/s  ({nbytes / 1e6:.0f} MB algorithmic)')


if __name__ == '__main__':
    main()
