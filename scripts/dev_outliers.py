"""Per-step times of eager training steps (HIP events between steps): prints the outliers.  usage: dev_outliers.py [C2|C4|C5] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tacotron_multispeaker_amd.engine import Engine
from tacotron_multispeaker_amd import synth
cfg = dict(C2=(32, 128, 640, 5, 0), C5=(16, 200, 800, 2, 460), C4=(32, 64, 480, 5, 460))[sys.argv[1] if len(sys.argv) > 1 else 'C4']
K = int(sys.argv[2]) if len(sys.argv) > 2 else 200
N, Ti, To, r, idn = cfg
eng = Engine(r=r, id_num=idn, seed=0)
pool = [synth.batch_to_device(synth.synth_batch(N, Ti, To, r, seed=1234 + i, id_num=idn), eng.dev) for i in range(4)]
static = [t.clone() if t is not None else None for t in pool[0]]
def load(i):
    if os.environ.get('STATIC'):
        for s_, t in zip(static, pool[i % 4]):
            if s_ is not None:
                s_.copy_(t)
        return static
    return pool[i % 4]
for i in range(5):
    eng.train_step(*load(i))
torch.cuda.synchronize()
import gc
if os.environ.get('NO_GC'):
    gc.collect(); gc.disable()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
if os.environ.get('WARM_EVENTS'):
    for e in ev:
        e.record()
    torch.cuda.synchronize()
host = []
ev[0].record()
for i in range(K):
    t0 = time.perf_counter()
    eng.train_step(*load(i))
    host.append((time.perf_counter() - t0) * 1e3)
    ev[i + 1].record()
torch.cuda.synchronize()
ms = np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(K)])
med = np.median(ms)
print('median %.3f ms  mean %.3f  max %.3f  err %d  host enqueue median %.2f ms max %.2f' % (med, ms.mean(), ms.max(), int(eng.err[0].item()), np.median(host), max(host)))
top = np.argsort(host)[-4:][::-1]
print('  largest host enqueue times:', ', '.join('step %d: %.1f ms' % (i, host[i]) for i in top))
for i in np.nonzero(ms > 1.2 * med)[0]:
    print('  step %d: %.3f ms (host enqueue %.2f ms)' % (i, ms[i], host[i]))
