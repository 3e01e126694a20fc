import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tacotron_multispeaker_amd._lib import lib, stream
x = torch.zeros(1 << 20, dtype=torch.int64, device='cuda'); err = torch.zeros(1, dtype=torch.int32, device='cuda'); sink = torch.zeros(4, device='cuda')
def run(nclus, cw, ln, same, sleep, threads, iters=2000):
    for _ in range(2): lib.taco_xchg_bench(x, err, sink, nclus, cw, ln, iters, same, sleep, threads, stream())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); lib.taco_xchg_bench(x, err, sink, nclus, cw, ln, iters, same, sleep, threads, stream()); e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
print('nclus cw len same sleep thr -> us/round', flush=True)
for (nclus, cw, ln, thr) in [(1, 2, 64, 256), (16, 4, 128, 256), (16, 8, 64, 512), (16, 8, 32, 512), (16, 8, 256, 512), (32, 8, 32, 512), (16, 2, 64, 256), (32, 4, 64, 256)]:
    for same in (1, 0):
        for sleep in (0, 1):
            print(nclus, cw, ln, same, sleep, thr, '-> %.2f' % run(nclus, cw, ln, same, sleep, thr), flush=True)
print('err', int(err.item()))
