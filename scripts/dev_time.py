import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import tacotron_np as onp
from tacotron_multispeaker_amd.engine import Engine

N, Ti, To, r = 32, 128, 640, 5
eng = Engine(r=r, seed=0)
b = onp.synth_batch(N, Ti, To, r, seed=1234)
dev = eng.dev
args = [torch.tensor(b[k], device=dev) for k in ('inputs', 'input_lengths', 'mel_targets', 'linear_targets')]
def step(): eng.train_step(*args)
def sync_time(fn, n):
    torch.cuda.synchronize(); t = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t) / n
step(); torch.cuda.synchronize()
print('loss', eng.loss_values(), flush=True)
print('eager step ms', sync_time(step, 3) * 1e3, flush=True)
# phase timing (eager)
def fwd(): eng.forward(args[0], args[1], args[2]); eng.loss(args[3])
print('fwd ms', sync_time(fwd, 3) * 1e3)
fwd()
dpos = eng._dpos
def bwd(): eng._dpos = dpos; eng.backward()
print('bwd ms', sync_time(bwd, 3) * 1e3)
print('opt ms', sync_time(eng.optimizer_step, 3) * 1e3)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step()
torch.cuda.synchronize()
print('graph step ms', sync_time(g.replay, 10) * 1e3, flush=True)
print('loss', eng.loss_values(), 'step', int(eng.global_step.item()))
