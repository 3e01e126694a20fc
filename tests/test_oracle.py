"""CPU tests of the oracle itself: the two independent restatements agree, analytic known-answer tests
(SURVEY.md Appendix A.12), finite-difference gradients, and the committed golden fixtures."""
import os

import numpy as np
import pytest
import torch

from oracle import tacotron_np as onp, tacotron_torch as ot

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def _cfg(name):
    g = np.load(os.path.join(GOLD, 'step_%s.npz' % name))
    N, Ti, To, r, idn, ps, bs = [int(x) for x in g['config']]
    P = onp.init_params(seed=ps, r=r, id_num=idn)
    b = onp.synth_batch(N, Ti, To, r, seed=bs, id_num=idn)
    return g, P, b, r, idn


@pytest.mark.parametrize('name', ['tiny', 'multi_r2'])
def test_numpy_and_torch_restatements_agree_and_match_golden(name):
    g, P, b, r, idn = _cfg(name)
    o = onp.forward(P, b['inputs'], b['input_lengths'], b['mel_targets'].astype(np.float64), b['identities'], idn, r)
    ts = ot.TrainState(P, torch.float64, id_num=idn, r=r)
    last = ts.forward_backward(b)
    for k in ('mel_outputs', 'linear_outputs', 'alignments'):
        t = last['out'][k].detach().numpy()
        assert np.abs(o[k] - t).max() < 1e-10                       # independent compositions, float64
        assert np.abs(t - g[k]).max() < 1e-6                        # committed fixture (float32 storage)
    l = onp.loss(o['mel_outputs'], o['linear_outputs'], b['mel_targets'], b['linear_targets'])
    assert abs(l[0] - last['loss']) < 1e-12 and abs(l[0] - g['loss'][0]) < 1e-9
    info = ts.apply(last)
    assert abs(info['global_norm'] - float(g['global_norm'])) < 1e-9
    for n, s in zip(g['param_names'], g['param_sum_after']):
        assert abs(float(ts.P[str(n)].sum()) - s) < 1e-7 * max(1.0, abs(s))


def test_finite_difference_gradients():
    """autograd gradients of the torch restatement == central differences of the NumPy restatement."""
    g, P, b, r, idn = _cfg('tiny')
    ts = ot.TrainState(P, torch.float64, r=r)
    grads = ts.forward_backward(b)['grads']

    def f(Pm):
        o = onp.forward(Pm, b['inputs'], b['input_lengths'], b['mel_targets'].astype(np.float64), None, 0, r)
        return onp.loss(o['mel_outputs'], o['linear_outputs'], b['mel_targets'], b['linear_targets'])[0]
    rng = np.random.RandomState(0)
    for name in ['attention/query_layer/kernel', 'encoder_cbhg/conv_bank/conv1d_4/kernel', 'decoder_gru_1/gates/kernel',
                 'post_cbhg/proj_1/gamma', 'attention_gru/candidate/bias', 'decoder_prenet/dense_1/kernel', 'linear/bias']:
        idx = tuple(rng.randint(0, s) for s in P[name].shape)
        eps = 1e-5
        Pp = dict(P); Pp[name] = P[name].copy(); Pp[name][idx] += eps
        Pm = dict(P); Pm[name] = P[name].copy(); Pm[name][idx] -= eps
        fd = (f(Pp) - f(Pm)) / (2 * eps)
        ag = float(grads[name][idx])
        assert abs(fd - ag) < 1e-6 + 1e-4 * abs(ag), (name, fd, ag)


# ---- Appendix A.12 known-answer tests ------------------------------------------------------------------
def test_conv_same_padding_shift():
    x = np.arange(2 * 6 * 1, dtype=np.float64).reshape(2, 6, 1)
    for k, shift in ((2, 0), (4, -1)):          # pad_left = (k-1)//2: one-hot at tap pad_left is the identity
        W = np.zeros((k, 1, 1)); W[(k - 1) // 2] = 1.0
        assert np.array_equal(onp.conv1d_same(x, W, np.zeros(1)), x)
        W = np.zeros((k, 1, 1)); W[k - 1] = 1.0                     # last tap reads x[t + k-1-pad_left]
        y = onp.conv1d_same(x, W, np.zeros(1))
        s = k - 1 - (k - 1) // 2
        assert np.array_equal(y[:, :-s], x[:, s:]) and np.all(y[:, -s:] == 0)
    xt = torch.tensor(x)
    for k in (2, 3, 4, 5):
        W = np.random.RandomState(k).randn(k, 1, 3)
        ref = torch.nn.functional.conv1d(xt.transpose(1, 2), torch.tensor(W).permute(2, 1, 0), padding='same').transpose(1, 2)
        assert np.abs(onp.conv1d_same(x, W, np.zeros(3)) - ref.numpy()).max() < 1e-12


def test_maxpool_last_frame_passthrough():
    x = np.array([[[1.], [5.], [2.], [7.]]])
    assert np.array_equal(onp.maxpool2_same(x)[0, :, 0], [5, 5, 7, 7])


def test_maxpool_first_max_routing_and_library_cross_check():
    """oracle max-pool: value and gradient equal F.max_pool1d on tie-free inputs; exact ties and ties within rounding
    noise (equal in exact arithmetic) send the gradient to the FIRST maximum like TF CPU MaxPoolGrad / the HIP kernel."""
    torch.manual_seed(3)
    x = torch.randn(3, 17, 5, dtype=torch.float64, requires_grad=True)
    g = torch.randn(3, 17, 5, dtype=torch.float64)
    ya = ot.maxpool2_same(x); ga, = torch.autograd.grad(ya, x, g)
    yb = ot.maxpool2_same_library(x); gb, = torch.autograd.grad(yb, x, g)
    assert torch.equal(ya, yb) and torch.equal(ga, gb)
    t = torch.tensor([[[1.0], [1.0], [1.0 + 1e-14], [1.0], [0.5]]], dtype=torch.float64, requires_grad=True)
    y = ot.maxpool2_same(t)
    d, = torch.autograd.grad(y, t, torch.tensor([[[1.0], [10.0], [100.0], [1000.0], [10000.0]]], dtype=torch.float64))
    # windows (0,1) tie -> 0; (1,2) near-tie -> 1; (2,3) near-tie -> 2; (3,4) -> 3; last frame passes through
    assert d[0, :, 0].tolist() == [1.0, 10.0, 100.0, 1000.0, 10000.0]
    assert y[0, 2, 0] == 1.0 + 1e-14                      # the VALUE is still the true maximum


def test_batch_norm_of_constant_channel_is_beta():
    x = np.ones((2, 5, 3)) * np.array([1.0, -2.0, 0.5])
    y, mu, var = onp.batch_norm_train(x, np.array([2.0, 3.0, 4.0]), np.array([0.1, 0.2, 0.3]))
    assert np.allclose(y, [0.1, 0.2, 0.3]) and np.allclose(var, 0)


def test_gru_zero_kernels():
    P = {'g/gates/kernel': np.zeros((6, 6)), 'g/gates/bias': np.ones(6), 'g/candidate/kernel': np.zeros((6, 3)),
         'g/candidate/bias': np.zeros(3)}
    h = np.array([[1.0, -2.0, 0.5]])
    assert np.allclose(onp.gru_cell(np.zeros((1, 3)), h, P, 'g'), onp.sigmoid(1.0) * h)     # 0.7311 h


def test_bigru_lengths():
    P = onp.init_params(seed=1, r=5)
    x = np.random.RandomState(2).randn(2, 6, 128)
    out = onp.bigru(x, np.array([6, 3]), P, 'encoder_cbhg')
    assert np.all(out[1, 3:] == 0)                                   # padded tail is zero in both directions
    single = onp.gru_cell(x[1:2, 2], np.zeros((1, 128)), P, 'encoder_cbhg/gru_bw')
    assert np.allclose(out[1, 2, 128:], single[0])                   # backward output at t = L-1: one step from zero


def test_attention_uniform_and_decoder_go_frame():
    P = onp.init_params(seed=1, r=5)
    enc = np.zeros((2, 7, 256))
    outs, aligns = onp.decoder_train(enc, np.random.RandomState(0).rand(2, 10, 80), P, 5)
    assert np.allclose(aligns, 1.0 / 7)                              # zero keys: uniform over ALL Ti positions
    outs2, _ = onp.decoder_train(enc, np.random.RandomState(1).rand(2, 10, 80), P, 5)
    assert np.allclose(outs[:, 0], outs2[:, 0])                      # step 0 sees the zero <GO> frame only
    assert not np.allclose(outs[:, 1], outs2[:, 1])


def test_loss_and_constants():
    a = np.random.RandomState(0).rand(2, 5, 80)
    b = np.random.RandomState(1).rand(2, 5, 1025)
    assert onp.loss(a, b, a, b)[0] == 0
    assert onp.n_priority_freq() == 307
    assert abs(onp.noam_lr(0.002, 0) - 5e-7) < 1e-15


def test_tf_adam_and_clip():
    P, G = {'w': np.array([1.0, 2.0])}, {'w': np.array([3.0, 4.0])}
    M, V = {'w': np.zeros(2)}, {'w': np.zeros(2)}
    norm = onp.adam_step(P, G, M, V, 1, 0.1)
    assert norm == 5.0
    g = np.array([0.6, 0.8])                                         # clipped to unit norm
    lr_t = 0.1 * np.sqrt(1 - 0.999) / (1 - 0.9)
    assert np.allclose(P['w'], np.array([1.0, 2.0]) - lr_t * (0.1 * g) / (np.sqrt(0.001 * g * g) + 1e-8))


def test_alignment_regularity_numpy_matches_torch_and_edge_cases():
    """tacotron.py:140-171 restated twice (numpy value, torch value + autograd); tf.slice edge cases of the
    `overwrought` term (needs >= 40 decoder steps; S == 40 selects nothing)."""
    import torch
    import pytest
    from oracle import tacotron_np as onp, tacotron_torch as ot
    rng = np.random.default_rng(3)
    a = rng.random((3, 9, 50))
    a /= a.sum(axis=1, keepdims=True)
    kw = dict(overwrought=0.3, oneorder_dynamic=0.2, variance_between_row=0.1, alignment_entropy=0.5)
    v_np = onp.alignment_regularity(a, **kw)
    v_t = float(ot.alignment_regularity(torch.tensor(a), **kw))
    assert abs(v_np - v_t) < 1e-12 * abs(v_np)
    assert onp.alignment_regularity(a) == 0.0                                   # all weights 0 (the reference default)
    # single terms against hand formulas
    e = np.exp(a - a.max(axis=2, keepdims=True)); pr = e / e.sum(axis=2, keepdims=True)
    assert abs(onp.alignment_regularity(a, overwrought=1.0) - pr[:, 0, 40:49].sum()) < 1e-12
    assert abs(onp.alignment_regularity(a, alignment_entropy=2.0) + 2.0 * np.mean(pr * np.log(pr))) < 1e-12
    a40 = a[:, :, :40]
    assert onp.alignment_regularity(a40, overwrought=1.0) == 0.0
    with pytest.raises(ValueError):
        onp.alignment_regularity(a[:, :, :39], overwrought=1.0)
    # autograd gradient of the torch restatement against central differences of the numpy one
    t = torch.tensor(a, requires_grad=True)
    ot.alignment_regularity(t, **kw).backward()
    g = t.grad.numpy()
    for idx in [(0, 0, 41), (1, 3, 0), (2, 8, 49), (0, 0, 10)]:
        d = 1e-6
        ap, am = a.copy(), a.copy()
        ap[idx] += d; am[idx] -= d
        fd = (onp.alignment_regularity(ap, **kw) - onp.alignment_regularity(am, **kw)) / (2 * d)
        assert abs(fd - g[idx]) < 1e-5 * max(1.0, abs(fd)), idx
