"""Step time as a function of the stream -> hardware-queue mapping: GPU_MAX_HW_QUEUES=<q> and SKIP=<k> torch pool streams burnt
before the engine creates its own (shifts which engine streams share an HSA queue).  One configuration per process."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tacotron_multispeaker_amd.engine import Engine
from tacotron_multispeaker_amd import synth
skip = int(os.environ.get('SKIP', '0'))
burn = [torch.cuda.Stream() for _ in range(skip)]
N, Ti, To, r = 32, 128, 640, 5
eng = Engine(r=r, seed=0)
args = synth.batch_to_device(synth.synth_batch(N, Ti, To, r, seed=1234), eng.dev)
for _ in range(6):
    eng.train_step(*args)
torch.cuda.synchronize()
ts = []
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(15):
        eng.train_step(*args)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / 15 * 1e3)
print('Q=%s SKIP=%d  step %.3f ms (min of 3x15; all %s)  err %s' % (os.environ.get('GPU_MAX_HW_QUEUES', 'default'), skip, min(ts),
      ' '.join('%.2f' % t for t in ts), eng.err.cpu().tolist()), flush=True)
