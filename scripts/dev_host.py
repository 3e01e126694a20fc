"""Host enqueue time of one eager training step vs its GPU time (is the step launch-bound?), and HIP-graph replay time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tacotron_multispeaker_amd.engine import Engine
from tacotron_multispeaker_amd import synth
N, Ti, To, r = 32, 128, 640, 5
eng = Engine(r=r, seed=0)
args = synth.batch_to_device(synth.synth_batch(N, Ti, To, r, seed=1234), eng.dev)
def step(): eng.train_step(*args[:4])
for _ in range(5): step()
torch.cuda.synchronize()
# host time: enqueue 3 steps behind a long-running blocker so the GPU never starves the measurement
x = torch.zeros(1 << 28, device=eng.dev)
for _ in range(4): x.add_(1.0)          # ~ms of queued work
t0 = time.perf_counter()
for _ in range(3): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
print('host enqueue ms/step %.3f' % ((t1 - t0) / 3 * 1e3))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize()
print('eager ms/step %.3f' % ((time.perf_counter() - t0) / 20 * 1e3))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step()
torch.cuda.synchronize()
for _ in range(3): g.replay()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): g.replay()
torch.cuda.synchronize()
print('graph ms/step %.3f   err %d' % ((time.perf_counter() - t0) / 20 * 1e3, int(eng.err[0].item())))
