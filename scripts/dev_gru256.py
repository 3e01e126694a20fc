"""Time the decoder GRU(256) cluster kernels alone (C2 shape N=32, S=128; one launch over all steps): us per decoder step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tacotron_multispeaker_amd._lib import lib, stream
torch.manual_seed(0)
dev = 'cuda'
N, S = int(os.environ.get('N', '32')), int(os.environ.get('S', '128'))
pad = int(os.environ.get('PAD', '150000'))
M = N * S
xp = torch.randn(M, 768, device=dev) * 0.3
whg, whc = torch.randn(256, 512, device=dev) * 0.06, torch.randn(256, 256, device=dev) * 0.06
res = torch.randn(M, 256, device=dev) * 0.3
t = [torch.zeros(M, 256, device=dev) for _ in range(5)]
d = torch.zeros(M, 256, device=dev)
xchg = torch.zeros(64 * 2048, dtype=torch.int64, device=dev)
err = torch.zeros(4, dtype=torch.int32, device=dev)
dout = torch.randn(M, 256, device=dev) * 0.1
dxp = torch.zeros(M, 768, device=dev)
carry = torch.zeros(N, 256, device=dev)
f = lambda: lib.taco_gru256_seq_fwd(xp, whg, whc, res, t[0], t[1], t[2], t[3], t[4], d, xchg, err, N, S, 0, S, pad, stream())
b = lambda: lib.taco_gru256_seq_bwd(dout, whg, whc, t[0], t[1], t[2], t[4], dxp, carry, xchg, err, N, S, 0, S, pad, stream())
big = torch.zeros(160 << 20, dtype=torch.float32, device=dev) if os.environ.get('COLD', '1') == '1' else None   # 640 MB > L2 + MALL
for name, fn, chk in (('fwd', f, d), ('bwd', b, dxp)):
    for _ in range(3):
        fn()
    tot = 0.0
    for _ in range(10):
        if big is not None:
            big.add_(1.0)                      # evicts the saved activations from L2 and the Infinity Cache, as the real step does
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        tot += e0.elapsed_time(e1)
    us = tot * 1e3 / 10
    print('N=%d S=%d %s: %.1f us per launch, %.3f us per step   checksum %.6f  err %s' % (N, S, name, us, us / S, float(chk.double().abs().sum()), err.tolist()), flush=True)
