"""attention recurrence kernels alone on the chip (C2): us per decoder step, forward and BPTT (weight gradients not overlapped)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['TACO_OVERLAP_WGRAD'] = '0'
import torch
from tacotron_multispeaker_amd.engine import Engine
from tacotron_multispeaker_amd import synth
cfg = dict(C2=(32, 128, 640, 5, 0), C5=(16, 200, 800, 2, 460), C4=(32, 64, 480, 5, 460))[os.environ.get('CFG', 'C2')]
N, Ti, To, r, idn = cfg
eng = Engine(r=r, id_num=idn, seed=0)
args = synth.batch_to_device(synth.synth_batch(N, Ti, To, r, seed=1234, id_num=idn), eng.dev)
for _ in range(3):
    eng.train_step(*args)
torch.cuda.synchronize()
eng.ktime = []
for _ in range(5):
    eng.train_step(*args)
torch.cuda.synchronize()
acc = {}
for name, flops, e0, e1 in eng.ktime:
    a = acc.setdefault(name, [0, 0.0]); a[0] += 1; a[1] += e0.elapsed_time(e1)
S = To // r
for k, (n, ms) in acc.items():
    if 'recurrence' in k or 'GRU' in k:
        steps = {'biGRU(128)': None}.get(k.split(' ')[0])
        print('%-52s %7.3f ms/step  (%d launches)%s' % (k, ms / 5, n / 5, '   %.2f us per decoder step' % (ms / 5 / S * 1e3) if 'recurrence' in k or '256' in k else ''), flush=True)
print('err', eng.err.cpu().tolist())
