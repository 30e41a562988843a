"""Would the backward pay for a second stream?  In rfn_decoder_bwd / rfn_prefix_bwd the weight-gradient products of a
recurrence (MFMA-bound, over all its steps at once) do not feed the next recurrence's backward chain (latency-bound, three
dependent launches per step): they could run beside it.  This probe runs a chain of dependent per-step products (the decoder's
Kb1 shape, variant picked by the library) on one stream and weight-gradient-sized GEMMs on another, and compares with running
them one after the other.    python tools/branch_overlap_probe.py [--batch 64]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import recurrent_fusion_network_amd._native as N  # noqa: E402

dev = torch.device('cuda:0')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--steps', type=int, default=25)
    ap.add_argument('--lean', action='store_true', help='weight-gradient GEMMs with GEMM_OPT_LDS_LEAN (64 KB of LDS per CU)')
    a = ap.parse_args()
    B, R, S = a.batch, 512, 17
    g = torch.Generator(device=dev).manual_seed(1)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=g) * 0.1  # noqa: E731
    ws = torch.empty(48 << 20, dtype=torch.uint8, device=dev)
    # the chain: Kb1-shaped products, each reading the previous one's output (dgates -> [dh | dz] -> ...): dependent launches
    dg = [rnd(B, 4 * R) for _ in range(2)]
    W = rnd(4 * R, 4 * R)           # [k][n]: 2048 -> 2048 so that the output feeds the next product
    chain_args = []
    for s in range(a.steps):
        src, dst = dg[s & 1], dg[(s + 1) & 1]
        outs = [dict(C=dst, ldc=4 * R, N=4 * R, accumulate=0, segs=[(src, 4 * R, W, 4 * R, 0, 4 * R, None)])]
        chain_args.append((N.cell_gemm_args(outs), outs))
    # the branch: weight gradients of a recurrence over all its steps: dW (2048 x 1536) = dgates^T (2048 x S*B) . X (S*B x 1536)
    dG, X = rnd(S * B, 4 * R), rnd(S * B, 3 * R)
    dWs = [torch.empty(4 * R, 3 * R, device=dev) for _ in range(4)]
    flags = N.GEMM_OPT_LDS_LEAN if a.lean else 0

    def chain():
        st = N.stream_ptr()
        for arr, _ in chain_args:
            N.check(N.lib.rfn_cell_gemm(B, 1, arr, R, 0.0, 0, 0, st))

    def branch():
        for dW in dWs:
            N.gemm(4 * R, 3 * R, [(dW, 3 * R, [(dG, 4 * R, 0, X, 3 * R, 0, S * B, None)])], ws=ws, flags=flags)

    side = torch.cuda.Stream()

    def timed(fn, reps=20):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps

    def both_serial():
        branch()
        chain()

    def both_overlapped():
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            branch()
        chain()
        main.wait_stream(side)

    tc, tb = timed(chain), timed(branch)
    ts, to = timed(both_serial), timed(both_overlapped)
    print('B = %d: chain of %d dependent Kb1-shaped products %.1f us | 4 weight-gradient GEMMs (2048 x 1536, K = %d)%s %.1f us | '
          'one after the other %.1f us | on two streams %.1f us' % (B, a.steps, tc, S * B, ' lean' if a.lean else '', tb, ts, to))


if __name__ == '__main__':
    main()
