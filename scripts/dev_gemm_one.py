"""Runs the dominant MFMA kernel (post_cbhg/proj_1 conv1d k=3 1024->256 over 20480 rows) 20 times; used under rocprofv3
--pmc to read its HBM traffic (FETCH_SIZE / WRITE_SIZE) for bench.py's roofline.traffic."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tacotron_multispeaker_amd._lib import lib, stream
M, T, cin, cout, kw = 20480, 640, 1024, 256, 3
x = torch.randn(M, cin, device='cuda'); w = torch.randn(kw, cin, cout, device='cuda') * 0.03
b = torch.randn(cout, device='cuda'); y = torch.empty(M, cout, device='cuda')
for _ in range(20):
    lib.taco_conv_gemm_fwd(x, w, b, y, M, T, cin, cout, kw, 0, cin, cout, cout, 1, 0, stream())
torch.cuda.synchronize()
