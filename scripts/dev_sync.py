"""Step time with a host synchronisation after every step (what train.py's loss read-back does) vs back-to-back steps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tacotron_multispeaker_amd.engine import Engine
from tacotron_multispeaker_amd import synth
N, Ti, To, r = 32, 128, 640, 5
eng = Engine(r=r, seed=0)
args = synth.batch_to_device(synth.synth_batch(N, Ti, To, r, seed=1234), eng.dev)
for _ in range(5):
    eng.train_step(*args)
torch.cuda.synchronize()
for mode in ('back-to-back', 'sync every step', 'sync + stage events', 'back-to-back', 'sync + stage events'):
    t0 = time.time()
    for _ in range(20):
        eng.sections = [] if 'events' in mode else None
        eng.train_step(*args)
        if mode != 'back-to-back':
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    print('%-18s %.3f ms/step  err %d' % (mode, (time.time() - t0) / 20 * 1e3, int(eng.err[0].item())), flush=True)
