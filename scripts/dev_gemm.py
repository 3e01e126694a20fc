"""Micro-benchmark of the conv/dense GEMM entry points on the C2 shapes (tuning aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tacotron_multispeaker_amd._lib import lib, stream
dev = 'cuda'
def timeit(fn, n=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
shapes = [  # name, M, T, Cin, Cout, kw, bank
    ('dense 16k', 16384, 16384, 2048, 2048, 1, 0), ('dense 8k', 8192, 8192, 4096, 4096, 1, 0),
    ('post proj_1', 20480, 640, 1024, 256, 3, 0), ('post bank', 20480, 640, 80, 1024, 8, 8), ('enc bank', 4096, 128, 128, 2048, 16, 16),
    ('enc proj_1', 4096, 128, 2048, 128, 3, 0), ('linear', 20480, 20480, 256, 1025, 1, 0), ('post xp', 20480, 20480, 128, 768, 1, 0),
    ('post hw', 20480, 20480, 128, 256, 1, 0), ('post proj_2', 20480, 640, 256, 80, 3, 0), ('dec xp', 4096, 4096, 256, 768, 1, 0)]
for name, M, T, cin, cout, kw, bank in shapes:
    ldw = 128 if bank else (cout + 3) & ~3
    taps = kw * (kw + 1) // 2 if bank else kw
    x = torch.randn(M, cin, device=dev); w = torch.randn(taps, cin, ldw, device=dev) * 0.05
    b = torch.randn(max(ldw, cout), device=dev); y = torch.empty(M, cout, device=dev)
    dy = torch.randn(M, (cout + 3) & ~3, device=dev); dx = torch.empty(M, cin, device=dev); dw = torch.zeros_like(w)
    lddy = (cout + 3) & ~3
    fl = 2.0 * M * taps * cin * (128 if bank else cout)
    t1 = timeit(lambda: lib.taco_conv_gemm_fwd(x, w, b, y, M, T, cin, cout, kw, bank, cin, ldw, cout, 1, 0, stream()))
    t2 = timeit(lambda: lib.taco_conv_gemm_bwd_data(dy, w, dx, M, T, cin, lddy if not bank else cout, kw, bank, lddy, ldw, cin, 0, stream()))
    t3 = timeit(lambda: lib.taco_conv_gemm_bwd_weight(x, dy, dw, M, T, cin, cout, kw, bank, cin, lddy, ldw, stream()))
    print('%-12s fwd %7.1f us %6.1f TF | dX %7.1f us %6.1f TF | dW %7.1f us %6.1f TF' % (name, t1, fl / t1 / 1e6, t2, fl / t2 / 1e6, t3, fl / t3 / 1e6), flush=True)
