"""bf16x3 products (conv_gemm_nt3) against exact-fp32 products on the input-gradient GEMMs of the model shapes: error against float64 and time.
usage: python scripts/dev_x3.py   (child processes: TACO_X3 is read once per process)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os
sys.path.insert(0, %r)
import numpy as np, torch
from tacotron_multispeaker_amd._lib import lib, stream
dev = 'cuda'
def conv_ref(x, w, T):
    M, cin = x.shape; kw = w.shape[0]; N = M // T
    xp = torch.nn.functional.pad(x.view(N, T, cin), (0, 0, (kw - 1) // 2, kw // 2))
    return sum(xp[:, j:j + T, :].reshape(M, cin) @ w[j] for j in range(kw))
def timeit(fn, n=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
shapes = [('post proj_1', 20480, 640, 1024, 256, 3, 0), ('linear', 20480, 20480, 256, 1025, 1, 0), ('enc bank', 4096, 128, 128, 2048, 16, 16),
          ('post bank', 20480, 640, 80, 1024, 8, 8), ('post xp', 20480, 20480, 128, 768, 1, 0), ('enc proj_1', 4096, 128, 2048, 128, 3, 0),
          ('post proj_2', 20480, 640, 256, 80, 3, 0), ('ragged', 4100, 205, 132, 260, 3, 0)]
for name, M, T, cin, cout, kw, bank in shapes:
    torch.manual_seed(1)
    ldw = 128 if bank else (cout + 3) & ~3
    taps = kw * (kw + 1) // 2 if bank else kw
    w = torch.randn(taps, cin, ldw, device=dev) / np.sqrt(cin * kw)
    lddy = cout if bank else (cout + 3) & ~3
    dy = torch.zeros(M, lddy, device=dev); dy[:, :cout] = torch.randn(M, cout, device=dev)
    dx = torch.empty(M, cin, device=dev)
    f = lambda: lib.taco_conv_gemm_bwd_data(dy, w, dx, M, T, cin, lddy if not bank else cout, kw, bank, lddy, ldw, cin, 0, stream())
    f(); torch.cuda.synchronize()
    # float64 reference: dX = sum_j shift(dY, -(j - pl)) . W_j^T
    d64, w64 = dy.double(), w.double()
    ref = torch.zeros(M, cin, dtype=torch.float64, device=dev)
    N = M // T
    k0 = 0
    for k in (range(1, kw + 1) if bank else [kw]):
        g = d64[:, (k - 1) * 128:k * 128] if bank else d64[:, :cout]
        wk = w64[k0:k0 + k, :, :128 if bank else cout]; k0 += k
        pl = (k - 1) // 2
        gp = torch.nn.functional.pad(g.view(N, T, -1), (0, 0, k // 2, pl))          # dX[t] = sum_j dY[t - (j - pl)] W_j^T
        for j in range(k):
            s = k // 2 - (j - pl)
            ref += gp[:, s:s + T, :].reshape(M, -1) @ wk[j].t()
    err = float((dx.double() - ref).norm() / ref.norm())
    mx = float((dx.double() - ref).abs().max() / ref.abs().max())
    fl = 2.0 * M * taps * cin * (128 if bank else cout)
    t = timeit(f)
    print('%%-12s X3=%%s  dX  rel-norm err %%.2e  max-abs/max %%.2e   %%7.1f us %%6.1f TF(fp32-equivalent)' %% (name, os.environ.get('TACO_X3', '1'), err, mx, t, fl / t / 1e6), flush=True)
    # forward: y = relu(conv(x, w) + b)
    x = torch.randn(M, cin, device=dev); b = torch.randn(max(ldw, cout) if not bank else cout, device=dev)
    y = torch.empty(M, cout, device=dev)
    g = lambda: lib.taco_conv_gemm_fwd(x, w, b, y, M, T, cin, cout, kw, bank, cin, ldw, cout, 1, 0, stream())
    g(); torch.cuda.synchronize()
    x64 = x.double()
    outs = []
    k0 = 0
    for k in (range(1, kw + 1) if bank else [kw]):
        wk = w64[k0:k0 + k, :, :128 if bank else cout]; k0 += k
        outs.append(conv_ref(x64, wk, T))
    ref = torch.relu(torch.cat(outs, 1) + b[:cout].double())
    err = float((y.double() - ref).norm() / ref.norm())
    mx = float((y.double() - ref).abs().max() / ref.abs().max())
    t = timeit(g)
    # weight gradient: dW_j = sum_m X[m + j - pl]^T dY[m]
    dw = torch.zeros_like(w)
    hfn = lambda: lib.taco_conv_gemm_bwd_weight(x, dy, dw, M, T, cin, cout, kw, bank, cin, lddy, ldw, stream())
    hfn(); torch.cuda.synchronize()
    refw = torch.zeros(taps, cin, 128 if bank else cout, dtype=torch.float64, device=dev)
    k0 = 0
    for k in (range(1, kw + 1) if bank else [kw]):
        g64 = d64[:, (k - 1) * 128:k * 128] if bank else d64[:, :cout]
        pl = (k - 1) // 2
        xp = torch.nn.functional.pad(x64.view(N, T, cin), (0, 0, pl, k // 2))
        for j in range(k):
            refw[k0 + j] = xp[:, j:j + T, :].reshape(M, cin).t() @ g64
        k0 += k
    got = dw[:, :, :128 if bank else cout].double()
    errw = float((got - refw).norm() / refw.norm())
    mxw = float((got - refw).abs().max() / refw.abs().max())
    def hrun():
        dw.zero_(); hfn()
    tw = timeit(hfn)
    print('%%-12s X3=%%s  dW  rel-norm err %%.2e  max-abs/max %%.2e   %%7.1f us %%6.1f TF(fp32-equivalent)' %% (name, os.environ.get('TACO_X3', '1'), errw, mxw, tw, fl / tw / 1e6), flush=True)
    print('%%-12s X3=%%s  fwd rel-norm err %%.2e  max-abs/max %%.2e   %%7.1f us %%6.1f TF(fp32-equivalent)' %% (name, os.environ.get('TACO_X3', '1'), err, mx, t, fl / t / 1e6), flush=True)
''' % ROOT
for x3 in ('0', '2'):
    subprocess.run([sys.executable, '-c', CHILD], env=dict(os.environ, TACO_X3=x3), check=False)
