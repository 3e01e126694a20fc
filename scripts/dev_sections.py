"""Un-profiled stage times of the eager C2 training step: HIP events on the main stream at the stage boundaries.
The last line compares the sum of the stages with the host wall clock of the same steps: trust the split only when they agree."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import collections
import torch
from tacotron_multispeaker_amd.engine import Engine
from tacotron_multispeaker_amd import synth
cfg = dict(C2=(32, 128, 640, 5, 0), C5=(16, 200, 800, 2, 460), C4=(32, 64, 480, 5, 460))[sys.argv[1] if len(sys.argv) > 1 else 'C2']
N, Ti, To, r, idn = cfg
eng = Engine(r=r, id_num=idn, seed=0)
args = synth.batch_to_device(synth.synth_batch(N, Ti, To, r, seed=1234, id_num=idn), eng.dev)
for _ in range(5):
    eng.train_step(*args)
torch.cuda.synchronize()
acc = collections.OrderedDict()
import time
steps = 10
wall = 0.0
for _ in range(steps):
    t0 = time.time()
    eng.sections = []
    eng.train_step(*args)
    torch.cuda.synchronize()
    wall += time.time() - t0
    ev = eng.sections
    for (n0, e0), (n1, e1) in zip(ev[:-1], ev[1:]):
        acc[n1] = acc.get(n1, 0.0) + e0.elapsed_time(e1)
tot = 0.0
for k, v in acc.items():
    print('%-20s %7.3f ms' % (k, v / steps)); tot += v / steps
print('%-20s %7.3f ms' % ('sum', tot))
print('%-20s %7.3f ms' % ('host wall clock', wall / steps * 1e3))
