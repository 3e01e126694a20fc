"""Host enqueue time and GPU time of the FIRST step after a device synchronisation (bench.py's timed region starts like that)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tacotron_multispeaker_amd.engine import Engine
from tacotron_multispeaker_amd import synth
N, Ti, To, r, idn = (32, 64, 480, 5, 460) if (len(sys.argv) > 1 and sys.argv[1] == 'C4') else (32, 128, 640, 5, 0)
eng = Engine(r=r, id_num=idn, seed=0)
args = synth.batch_to_device(synth.synth_batch(N, Ti, To, r, seed=1234, id_num=idn), eng.dev)
for _ in range(8):
    eng.train_step(*args)
torch.cuda.synchronize()
res = []
for k in range(30):
    if os.environ.get('SLEEP'):
        time.sleep(float(os.environ['SLEEP']))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    eng.train_step(*args)
    e1.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    res.append(((t1 - t0) * 1e3, e0.elapsed_time(e1)))
h = np.array([a for a, _ in res]); g = np.array([b for _, b in res])
print('first step after a sync, 30 trials: host enqueue median %.2f max %.2f ms | GPU median %.3f max %.3f ms' % (np.median(h), h.max(), np.median(g), g.max()))
print('  GPU times:', ' '.join('%.1f' % x for x in g))
