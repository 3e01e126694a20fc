"""Time the biGRU(128) recurrence kernels alone (post-net shape N=32, T=640 and encoder shape T=128): us per step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tacotron_multispeaker_amd._lib import lib, stream
torch.manual_seed(0)
dev = 'cuda'
for (N, T) in ((32, 640), (32, 128)):
    M = N * T
    xp = torch.randn(M, 768, device=dev) * 0.3
    W = lambda a, b: torch.randn(a, b, device=dev) * 0.08
    wgf, wcf, wgb, wcb = W(128, 256), W(128, 128), W(128, 256), W(128, 128)
    out = torch.zeros(M, 256, device=dev); ruc = torch.zeros(2, N, T, 384, device=dev)
    dout = torch.randn(M, 256, device=dev) * 0.1; dxp = torch.zeros(M, 768, device=dev)
    hp = torch.zeros(2, M, 128, device=dev); rh = torch.zeros(2, M, 128, device=dev)
    state = torch.zeros(2, N, 128, device=dev)
    f = lambda: lib.taco_gru128_seq_fwd(xp, 768, wgf, wcf, wgb, wcb, None, out, 256, ruc, N, T, 2, 0, T, state, 0, stream())
    b = lambda: lib.taco_gru128_seq_bwd(dout, 256, wgf, wcf, wgb, wcb, None, out, 256, ruc, dxp, 768, hp, rh, N, T, 2, 0, T, state, 0, stream())
    for name, fn in (('fwd', f), ('bwd', b)):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record(); e1.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 10
        print('N=%d T=%d %s: %.1f us per launch, %.3f us per step   checksum %.6f' % (N, T, name, us, us / T, float((out if name == 'fwd' else dxp).double().abs().sum())), flush=True)
