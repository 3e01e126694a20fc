"""Where do HIP and the float64 oracle disagree in the encoder CBHG backward on zero-padded text?  (developer diagnostic)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import tacotron_np as onp, tacotron_torch as ot
from tacotron_multispeaker_amd.engine import Engine
N, Ti, To, r, idn = [int(x) for x in (sys.argv[1:6] if len(sys.argv) > 5 else (2, 160, 40, 5, 0))]
P = onp.init_params(seed=33, r=r, id_num=idn)
rng = np.random.RandomState(6)
for k in P:
    if k.endswith(('/bias', '/beta')):
        P[k] = P[k] + 0.1 * rng.standard_normal(P[k].shape)
b = onp.synth_batch(N, Ti, To, r, seed=43, id_num=idn)
print('lengths', b['input_lengths'])
ot.DEBUG_TAPS = {}
ts = ot.TrainState(P, torch.float64, id_num=idn, r=r)
last = ts.forward_backward(b)
taps = ot.DEBUG_TAPS
eng = Engine(id_num=idn, r=r, named_params=P)
dev = eng.dev
t = lambda k, dt: torch.tensor(b[k], device=dev, dtype=dt) if b.get(k) is not None else None
eng.forward(t('inputs', torch.int32), t('input_lengths', torch.int32), t('mel_targets', torch.float32), t('identities', torch.int32))
eng.loss(t('linear_targets', torch.float32))
eng.backward()
torch.cuda.synchronize()
eng.check_errors()
sc = 'encoder_cbhg'
C = 2048
# oracle bank is post-BN (conv->relu->BN); HIP 'bank' buffer is pre-BN (post-relu) -> compare pooled tensors and gradients
hp = eng._bufs[sc + '/pool'].cpu().numpy().reshape(N, Ti, C).astype(np.float64)
op = taps[sc + '/pooled'].detach().numpy()
print('pooled rel err', np.abs(hp - op).max() / np.abs(op).max())
hdp = eng._bufs[sc + '/dpool'].cpu().numpy().reshape(N, Ti, C).astype(np.float64)
odp = taps[sc + '/pooled'].grad.numpy()
print('dpooled rel L2 err', np.sqrt(((hdp - odp) ** 2).sum() / (odp ** 2).sum()))
# gradient wrt the BN output (oracle) -- HIP does not materialise it; rebuild from HIP dpool with first-max routing in fp64
ob = taps[sc + '/bank'].detach().numpy()
odb = taps[sc + '/bank'].grad.numpy()
nxt = np.concatenate([ob[:, 1:], np.full_like(ob[:, :1], -np.inf)], 1)
first = ob >= nxt - 1e-11 * np.abs(ob).max()
exact_first = ob >= nxt
print('near-ties decided by tolerance (oracle):', int((first != exact_first).sum()), 'of', first.size, '; exact ties', int((ob == nxt).sum()))
g1 = np.where(first, odp, 0.0); g2 = odp - g1
mine = g1.copy(); mine[:, 1:] += g2[:, :-1]
print('oracle dbank vs manual routing:', np.abs(mine - odb).max())
# HIP: bank (pre-BN) + scale/shift -> b; route HIP dpool with HIP's exact comparisons
hb = eng._bufs[sc + '/bank'].cpu().numpy().reshape(N, Ti, C)
s_, h_ = eng._bufs[sc + '/conv_bank/bn_scale'].cpu().numpy(), eng._bufs[sc + '/conv_bank/bn_shift'].cpu().numpy()
hbn = (hb * s_ + h_).astype(np.float32)      # not fma-exact, but ties stay ties
hn = np.concatenate([hbn[:, 1:], np.full_like(hbn[:, :1], -np.inf)], 1)
hfirst = hbn >= hn
dis = hfirst != first
print('routing decisions that differ HIP vs oracle:', int(dis.sum()))
idx = np.argwhere(dis)[:20]
for n, tt, c in idx:
    print('  n %d t %d (len %d) c %d (block k=%d): HIP b[t]=%.9g b[t+1]=%.9g | oracle %.12g %.12g | dpool %.3e' % (
        n, tt, b['input_lengths'][n], c, c // 128 + 1, hbn[n, tt, c], hn[n, tt, c], ob[n, tt, c], nxt[n, tt, c], odp[n, tt, c]))
hdb = eng._bufs[sc + '/dbank'].cpu().numpy().reshape(N, Ti, C).astype(np.float64)     # gradient wrt conv pre-activation (after BN bwd + relu)
print('per bank block: relative L2 error of the kernel gradient')
grads = eng.export_named('grads')
for k in range(1, 17):
    nm = '%s/conv_bank/conv1d_%d/kernel' % (sc, k)
    v = last['grads'][nm].numpy()
    print('  k=%2d  %.2e   decisions differing in block: %d' % (k, np.sqrt(((grads[nm] - v) ** 2).sum() / (v ** 2).sum()), int(dis[:, :, (k - 1) * 128:k * 128].sum())))
