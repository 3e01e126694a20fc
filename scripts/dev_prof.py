import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import tacotron_np as onp
from tacotron_multispeaker_amd.engine import Engine
N, Ti, To, r = 32, 128, 640, 5
eng = Engine(r=r, seed=0)
b = onp.synth_batch(N, Ti, To, r, seed=1234)
args = [torch.tensor(b[k], device=eng.dev) for k in ('inputs', 'input_lengths', 'mel_targets', 'linear_targets')]
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    eng.train_step(*args)
torch.cuda.synchronize()
