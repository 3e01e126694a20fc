"""Few steps of one or two engines under rocprofv3 --kernel-trace: which HSA queue does each kernel family land on?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tacotron_multispeaker_amd.engine import Engine
from tacotron_multispeaker_amd import synth
N, Ti, To, r = 32, 128, 640, 5
for k in range(int(os.environ.get('ENGINES', '2'))):
    eng = Engine(r=r, seed=0)
    args = synth.batch_to_device(synth.synth_batch(N, Ti, To, r, seed=1234), eng.dev)
    for _ in range(3):
        eng.train_step(*args)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        eng.train_step(*args)
    torch.cuda.synchronize()
    print('engine %d: %.3f ms/step' % (k, (time.perf_counter() - t0) / 10 * 1e3), flush=True)
    lib_marker = torch.zeros(1, device='cuda'); lib_marker += float(k + 1)     # a torch kernel marks the boundary in the trace
    torch.cuda.synchronize()
