"""Generates tests/golden/step_*.npz from the float64 oracle (oracle/tacotron_torch.py, cross-checked against
oracle/tacotron_np.py): one full training step on small seeded configs.  The reference itself cannot be run
(TensorFlow 1.x absent) and holds no golden vectors, so these fixtures pin the ORACLE (regression guard) and
give the GPU tests a device-independent target; see DESIGN.md 'parity unpinned'."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import tacotron_np as onp, tacotron_torch as ot  # noqa: E402

CONFIGS = {
    'tiny': dict(N=2, Ti=12, To=20, r=5, id_num=0, pseed=3, bseed=11),
    'multi_r2': dict(N=3, Ti=21, To=24, r=2, id_num=4, pseed=4, bseed=12),
}


def make(name, c):
    P = onp.init_params(seed=c['pseed'], r=c['r'], id_num=c['id_num'])
    b = onp.synth_batch(c['N'], c['Ti'], c['To'], c['r'], seed=c['bseed'], id_num=c['id_num'])
    o_np = onp.forward(P, b['inputs'], b['input_lengths'], b['mel_targets'].astype(np.float64), b['identities'],
                       c['id_num'], c['r'])
    ts = ot.TrainState(P, torch.float64, id_num=c['id_num'], r=c['r'])
    last = ts.forward_backward(b)
    assert np.abs(o_np['mel_outputs'] - last['out']['mel_outputs'].detach().numpy()).max() < 1e-10
    info = ts.apply(last)
    res = dict(mel_outputs=last['out']['mel_outputs'].detach().numpy().astype(np.float32),
               linear_outputs=last['out']['linear_outputs'].detach().numpy().astype(np.float32),
               alignments=last['out']['alignments'].detach().numpy().astype(np.float32),
               loss=np.array([last['loss'], last['mel_loss'], last['linear_loss']]),
               global_norm=np.array(info['global_norm']), learning_rate=np.array(info['learning_rate']))
    names = list(last['grads'].keys())
    res['grad_names'] = np.array(names)
    res['grad_l2'] = np.array([float(last['grads'][k].norm()) for k in names])
    res['grad_sum'] = np.array([float(last['grads'][k].sum()) for k in names])
    res['param_names'] = np.array(list(ts.P.keys()))
    res['param_sum_after'] = np.array([float(v.sum()) for v in ts.P.values()])
    res['param_abs_after'] = np.array([float(v.abs().sum()) for v in ts.P.values()])
    res['config'] = np.array([c['N'], c['Ti'], c['To'], c['r'], c['id_num'], c['pseed'], c['bseed']])
    out = os.path.join(HERE, 'step_%s.npz' % name)
    np.savez_compressed(out, **res)
    print(name, 'loss', res['loss'], 'gnorm', res['global_norm'], os.path.getsize(out))


if __name__ == '__main__':
    import warnings
    warnings.filterwarnings('ignore')
    for n, c in CONFIGS.items():
        make(n, c)
