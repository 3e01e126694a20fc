"""Host enqueue time vs GPU time of the eager training step (is the step ever host-bound?)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import tacotron_np as onp
from tacotron_multispeaker_amd.engine import Engine
N, Ti, To, r = 32, 128, 640, 5
eng = Engine(r=r, seed=0)
b = onp.synth_batch(N, Ti, To, r, seed=1234)
args = [torch.tensor(b[k], device=eng.dev) for k in ('inputs', 'input_lengths', 'mel_targets', 'linear_targets')]
for _ in range(3): eng.train_step(*args)
torch.cuda.synchronize()
n = 10
t0 = time.time()
for _ in range(n): eng.train_step(*args)
t1 = time.time()
torch.cuda.synchronize()
t2 = time.time()
print('host enqueue ms/step %.2f   total ms/step %.2f' % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(3): eng.train_step(*args)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
