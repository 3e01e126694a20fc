"""Prints the worst relative-L2 gradient error per oracle test configuration (how far the parity tests sit from their bars)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
import test_gpu_parity as T
from oracle import tacotron_np as onp, tacotron_torch as ot
cfgs = [(4, 48, 120, 5, 0, 0), (5, 17, 35, 5, 0, 0), (1, 9, 12, 3, 0, 0), (3, 33, 16, 1, 7, 0), (2, 40, 64, 2, 3, 0), (4, 24, 40, 5, 0, 1), (4, 48, 120, 5, 0, 2)]
for cfg in cfgs:
    N, Ti, To, r, idn, mode = cfg
    P = onp.init_params(seed=21, r=r, id_num=idn)
    rng = np.random.RandomState(5)
    for k in P:
        if k.endswith(('/bias', '/beta')): P[k] = P[k] + 0.1 * rng.standard_normal(P[k].shape)
        if k.endswith('/gamma'): P[k] = P[k] * (1 + 0.2 * rng.standard_normal(P[k].shape))
    b = onp.synth_batch(N, Ti, To, r, seed=31, id_num=idn)
    if mode == 0:
        pad = b['inputs'] == 0
        b['inputs'][pad] = np.random.RandomState(7).randint(2, 7352, size=int(pad.sum()))
    if mode == 1:
        b['input_lengths'][0] = 1; b['inputs'][0, :] = 0; b['inputs'][0, 0] = 1
    ts = ot.TrainState(P, torch.float64, id_num=idn, r=r)
    last = ts.forward_backward(b)
    worst = []
    for rep in range(3):
        o = T.run_engine_step(P, b, r, idn, apply=False)
        gmax = max(float(v.norm()) for v in last['grads'].values())
        w = max(((np.sqrt(((o['grads'][k] - v.numpy()) ** 2).sum()) - 1e-6 * gmax) / max(np.sqrt((v.numpy() ** 2).sum()), 1e-30), k) for k, v in last['grads'].items())
        worst.append(w)
    print(cfg, ' '.join('%.2e(%s)' % (w[0], w[1].split('/')[-2] if '/' in w[1] else w[1]) for w in worst), 'fwd mel rel %.1e' % T.rel(o['mel'], last['out']['mel_outputs'].detach().numpy()), flush=True)
