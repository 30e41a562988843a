"""K-split choice of the medium big-tile GEMMs: time per forced split (RFN_GEMM_OPT_FORCE_SPLIT) next to the
library's own choice, at the shapes the train step launches.   python tools/split_probe.py [c3|c2|het]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from recurrent_fusion_network_amd import _native as nv  # noqa: E402

SHAPES = {
    'c3': [
        ('logit dX   4352x512 K=9488 NN', 4352, 512, [9488], 1, 0, 1),
        ('logit dW   9472x512 K=4352 TN', 9472, 512, [4352], 0, 0, 1),
        ('stage-I gates 4x 256x2048 K=2048+2048 NT', 256, 2048, [2048, 2048], 1, 1, 4),
        ('stage-I dH 4x 256x2048 K=2048 NN', 256, 2048, [2048], 1, 0, 4),
        ('i2h fwd 4352x2048 K=512 NT', 4352, 2048, [512], 1, 1, 1),
        ('decoder dW 2048x512 K=4352 TN', 2048, 512, [4352], 0, 0, 1),
        ('decoder dW 2048x2048 K=4352 TN', 2048, 2048, [4352], 0, 0, 1),
    ],
    'het': [
        ('enc1 dW 4096x1536 K=16384 TN', 4096, 1536, [16384], 0, 0, 1),
        ('enc3 dW 4096x2176 K=12544 TN', 4096, 2176, [12544], 0, 0, 1),
        ('enc2 dW 4096x1280 K=16384 TN', 4096, 1280, [16384], 0, 0, 1),
    ],
    'c2': [
        ('logit dX   1088x512 K=9488 NN', 1088, 512, [9488], 1, 0, 1),
        ('logit dW   9472x512 K=1088 TN', 9472, 512, [1088], 0, 0, 1),
        ('decoder dW 2048x512 K=1088 TN', 2048, 512, [1088], 0, 0, 1),
    ],
}


def run(name, M, N, Ks, ak, bk, ng, ws):
    dev = 'cuda'
    probs = []
    keep = []
    for g in range(ng):
        C = torch.empty(M, N, device=dev)
        segs = []
        for K in Ks:
            A = torch.randn((M, K) if ak else (K, M), device=dev)
            B = torch.randn((N, K) if bk else (K, N), device=dev)
            keep += [A, B]
            segs.append((A, K if ak else M, ak, B, K if bk else N, bk, K, None))
        probs.append((C, N, segs))
        keep.append(C)
    flops = 2.0 * M * N * sum(Ks) * ng
    out = []
    for split in [0] + list(range(1, 17)):
        flags = (split & 31) << 8
        for _ in range(10):
            nv.gemm(M, N, probs, ws=ws, flags=flags)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            nv.gemm(M, N, probs, ws=ws, flags=flags)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / 50
        out.append('%s %.0f' % ('auto' if split == 0 else '%d:' % split, us))
    print(name, 'tiles', -(-M // 128) * -(-N // 128) * ng, 'iters', sum(-(-K // 32) for K in Ks), '| us:', ' '.join(out), flush=True)


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else 'c3'
    ws = torch.empty(int(os.environ.get("PROBE_WS_MIB", "256")) << 20, dtype=torch.uint8, device='cuda')
    for sh in SHAPES[which]:
        run(*sh, ws)


if __name__ == '__main__':
    main()
