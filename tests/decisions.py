"""Decision alignment between the HIP path and the float64 oracle (test infrastructure).

The training step is piecewise smooth.  A ReLU pre-activation or a max-pool operand pair that agrees with the kink to
within fp32 rounding may fall on the other side in ANY fp32 implementation, and one such unit moves a small gradient
tensor by a few 1e-3 of its norm (traced: N=2, T_in=160, one pooling window with operands 1.3e-6 apart moved
encoder_cbhg/conv_bank/conv1d_12/kernel by 3.2e-3 while the other 15 bank kernels agreed to 3e-6).  Instead of widening
the gradient tolerance, the tests
  1. let the oracle record its own decisions and their margins (oracle/tacotron_torch.py DECISIONS),
  2. read the HIP path's decisions back from the activations it saved for its own backward,
  3. require every differing decision to be a genuine near-tie in float64 (margin <= NEAR x the site's RMS) and their
     count to stay at the rounding-noise rate (<= 4 + RATE x decisions), and
  4. compare all gradients at 1e-3 per tensor against the oracle evaluated on the SAME linear piece (replay).
"""
import numpy as np
import torch

from oracle import tacotron_torch as ot

NEAR = 3e-5        # a flipped decision's float64 margin, relative to the RMS of its site: fp32 rounding of a K<=6144 dot product
RATE = 2e-5        # flips per decision (expected ~3e-6: 2 x relative rounding x density of margins at 0)


def hip_decisions(eng):
    """site -> bool tensor (CPU, channel-last), from the buffers the HIP path saved for backward."""
    N, Ti, To, S = eng.dims
    b = eng._bufs
    out = {'prenet/dense_1': (b['enc_p1'] > 0).view(N, Ti, -1).cpu(), 'prenet/dense_2': (b['enc_p2'] > 0).view(N, Ti, -1).cpu()}
    for sc, T, K in (('encoder_cbhg', Ti, 16), ('post_cbhg', To, 8)):
        bank = b[sc + '/bank'].view(N, T, K * 128)
        on = (bank > 0).cpu()
        for k in range(1, K + 1):
            out['%s/conv_bank/conv1d_%d' % (sc, k)] = on[:, :, (k - 1) * 128:k * 128]
        # the kernels pool fma(x, scale, shift) in fp32: the product of two floats is exact in float64, one rounding to fp32
        bb = (bank.double() * b[sc + '/conv_bank/bn_scale'].double() + b[sc + '/conv_bank/bn_shift'].double()).float()
        first = torch.ones_like(bb, dtype=torch.bool)
        first[:, :-1] = bb[:, :-1] >= bb[:, 1:]
        out[sc + '/pool'] = first.cpu()
        out[sc + '/proj_1'] = (b[sc + '/c1'] > 0).view(N, T, -1).cpu()
        for i in range(1, 5):
            out['%s/highway_%d/H' % (sc, i)] = (b['%s/hwZ%d' % (sc, i)][:, :128] > 0).view(N, T, 128).cpu()
    p1, p2 = (b['P1'] > 0).view(N, S, 256).cpu(), (b['P2'] > 0).view(N, S, 128).cpu()
    for s in range(S):
        out['decoder_prenet/dense_1@%d' % s] = p1[:, s]
        out['decoder_prenet/dense_2@%d' % s] = p2[:, s]
    return out


def compare(rec, hip):
    """-> (number of differing decisions, total decisions); asserts that every differing one is a float64 near-tie."""
    assert set(rec['masks']) == set(hip), set(rec['masks']) ^ set(hip)
    flips = total = 0
    for site, m in hip.items():
        o = rec['masks'][site]
        assert o.shape == m.shape, (site, o.shape, m.shape)
        if site.endswith('/pool'):
            o = o.clone(); o[:, -1] = True
        diff = o != m
        total += diff.numel()
        n = int(diff.sum())
        if n:
            mg = rec['margins'][site]
            fin = mg[torch.isfinite(mg)]
            rms = float(torch.sqrt((fin ** 2).mean()))
            worst = float(mg[diff].max())
            assert worst <= NEAR * rms, 'HIP took a decision the float64 oracle calls clear: %s margin %.3e (site rms %.3e)' % (site, worst, rms)
            flips += n
    assert flips <= 4 + RATE * total, '%d of %d decisions differ: more than fp32 rounding explains' % (flips, total)
    return flips, total


def oracle_step_aligned(P, b, r, idn, run_engine, regularity=None):
    """run_engine() -> (engine, outputs dict).  Returns (TrainState, last, outputs, flips): `last` is the oracle's
    forward/backward on the linear piece the HIP path took (identical to the plain oracle when flips == 0)."""
    ot.DECISIONS = dict(mode='record', masks={}, margins={})
    try:
        ts = ot.TrainState(P, torch.float64, id_num=idn, r=r, regularity=regularity)
        last = ts.forward_backward(b)
        rec = ot.DECISIONS
    finally:
        ot.DECISIONS = None
    eng, o = run_engine()
    hip = hip_decisions(eng)
    flips, total = compare(rec, hip)
    if flips:
        ot.DECISIONS = dict(mode='replay', masks=hip)
        try:
            ts = ot.TrainState(P, torch.float64, id_num=idn, r=r, regularity=regularity)
            last = ts.forward_backward(b)
        finally:
            ot.DECISIONS = None
    print('decisions: %d of %d differ from the float64 oracle (all within fp32 rounding of the kink)' % (flips, total))
    return ts, last, o, flips
