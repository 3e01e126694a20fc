"""us per cluster all-gather round: agent-scope granules vs same-XCD L2 granules (csrc/xcd_granule.hpp)."""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tacotron_multispeaker_amd._lib import lib, stream
dll = lib.load()
fn = dll.taco_dev_xchg_bench
fn.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 8 + [ctypes.c_void_p]
fn.restype = ctypes.c_int
x = torch.zeros(1 << 20, dtype=torch.int64, device='cuda'); err = torch.zeros(1, dtype=torch.int32, device='cuda'); sink = torch.zeros(4, device='cuda')


def run(nclus, cw, ln, same, sleep, threads, mode, iters=2000):
    call = lambda: fn(x.data_ptr(), err.data_ptr(), sink.data_ptr(), nclus, cw, ln, iters, same, sleep, threads, mode, stream())
    for _ in range(2):
        assert call() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); call(); e1.record(); e1.synchronize()
    e = int(err.item()); err.zero_()
    return e0.elapsed_time(e1) * 1e3 / iters, e


print('nclus cw len same_xcd_map sleep thr mode -> us/round (err: 0 ok, 1 timeout, 2 cluster not on one XCD, 3 wrong word)', flush=True)
for (nclus, cw, ln, thr) in [(16, 8, 64, 512), (16, 8, 32, 512), (16, 8, 256, 512), (16, 4, 128, 256), (32, 4, 64, 256), (8, 8, 64, 512)]:
    for same in (1, 0):
        for mode in (0, 1):
            t, e = run(nclus, cw, ln, same, 0, thr, mode)
            print(nclus, cw, ln, same, 0, thr, mode, '-> %.2f us  err %d' % (t, e), flush=True)
