"""ORACLE (test infrastructure, NOT product code) -- float64 NumPy restatement of the
Tacotron multispeaker training-step forward pass + loss of Jim-Song/tacotron_multispeaker.

PARITY UNPINNED: the reference delegates all arithmetic to TensorFlow 1.3/1.4
(README.md:35), which is neither vendored in /root/reference nor installable here, and the
reference holds no tests / golden vectors.  This file therefore
restates the *published* TF-1.4 operator semantics at the
reference's own call sites.  It is pinned only by (1) the independent PyTorch composition
in oracle/tacotron_torch.py (agreement <= 1e-10 in float64), (2) analytic known-answer
tests (tests/test_oracle_known_answers.py) and (3) finite-difference gradient checks.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Each function cites the reference file:line whose behaviour it follows.
"""
from collections import OrderedDict

import numpy as np

NUM_SYMBOLS2 = 7352          # len(text.symbols.symbols2); reference models/tacotron.py:40
BN_EPS = 1e-3                # tf.layers.batch_normalization default epsilon
BN_MOMENTUM = 0.99           # tf.layers.batch_normalization default momentum


# --------------------------------------------------------------------------------------
# Parameter inventory (creation order of Tacotron.initialize)
# --------------------------------------------------------------------------------------
def _glorot_uniform(rng, shape):
    """TF default get_variable initializer (glorot_uniform). Fans as TF _compute_fans."""
    if len(shape) == 1:
        fan_in = fan_out = shape[0]
    elif len(shape) == 2:
        fan_in, fan_out = shape
    else:
        rf = int(np.prod(shape[:-2]))
        fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
    limit = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-limit, limit, size=shape)


def _trunc_normal(rng, shape, stddev):
    """tf.truncated_normal_initializer: re-draw samples beyond 2 sigma (tacotron.py:44,51)."""
    x = rng.normal(0.0, 1.0, size=shape)
    bad = np.abs(x) > 2.0
    while bad.any():
        x[bad] = rng.normal(0.0, 1.0, size=int(bad.sum()))
        bad = np.abs(x) > 2.0
    return x * stddev


def _conv_bn(P, rng, scope, k, cin, cout):
    """modules.py:93-101  conv1d kernel/bias + batch-norm gamma/beta/moving stats."""
    P[scope + '/kernel'] = _glorot_uniform(rng, (k, cin, cout))
    P[scope + '/bias'] = np.zeros(cout)
    P[scope + '/gamma'] = np.ones(cout)
    P[scope + '/beta'] = np.zeros(cout)
    P[scope + '/moving_mean'] = np.zeros(cout)
    P[scope + '/moving_variance'] = np.ones(cout)


def _gru(P, rng, scope, n_in, n):
    """tf.contrib.rnn.GRUCell variables (Appendix A.5): gates bias initialised to 1.0."""
    P[scope + '/gates/kernel'] = _glorot_uniform(rng, (n_in + n, 2 * n))
    P[scope + '/gates/bias'] = np.ones(2 * n)
    P[scope + '/candidate/kernel'] = _glorot_uniform(rng, (n_in + n, n))
    P[scope + '/candidate/bias'] = np.zeros(n)


def _cbhg(P, rng, scope, K, cin, projections):
    """modules.py:35-74 variables in creation order."""
    for k in range(1, K + 1):
        _conv_bn(P, rng, '%s/conv_bank/conv1d_%d' % (scope, k), k, cin, 128)
    _conv_bn(P, rng, scope + '/proj_1', 3, K * 128, projections[0])
    _conv_bn(P, rng, scope + '/proj_2', 3, projections[0], projections[1])
    if projections[1] != 128:  # modules.py:59-60
        P[scope + '/highway_dense/kernel'] = _glorot_uniform(rng, (projections[1], 128))
        P[scope + '/highway_dense/bias'] = np.zeros(128)
    for i in range(1, 5):      # modules.py:77-90
        P['%s/highway_%d/H/kernel' % (scope, i)] = _glorot_uniform(rng, (128, 128))
        P['%s/highway_%d/H/bias' % (scope, i)] = np.zeros(128)
        P['%s/highway_%d/T/kernel' % (scope, i)] = _glorot_uniform(rng, (128, 128))
        P['%s/highway_%d/T/bias' % (scope, i)] = np.full(128, -1.0)
    _gru(P, rng, scope + '/gru_fw', 128, 128)
    _gru(P, rng, scope + '/gru_bw', 128, 128)


NON_TRAINABLE_SUFFIXES = ('/moving_mean', '/moving_variance')


def is_trainable(name):
    return not name.endswith(NON_TRAINABLE_SUFFIXES)


def init_params(seed=0, r=5, id_num=0, num_mels=80, num_freq=1025, vocab=NUM_SYMBOLS2,
                embed_text=256, embed_id=64):
    """All variables of models/tacotron.py:35-104 with TF initialisers (Appendix A.9/B),
    drawn in creation order from one np.random.RandomState(seed).  float64."""
    rng = np.random.RandomState(seed)
    P = OrderedDict()
    P['embedding'] = _trunc_normal(rng, (vocab, embed_text), 0.5)         # tacotron.py:42-44
    e_in = embed_text
    if id_num > 1:                                                         # tacotron.py:48-51
        P['embedding_id'] = _trunc_normal(rng, (id_num, embed_id), 0.5)
        e_in += embed_id
    P['prenet/dense_1/kernel'] = _glorot_uniform(rng, (e_in, 256))        # modules.py:5-12
    P['prenet/dense_1/bias'] = np.zeros(256)
    P['prenet/dense_2/kernel'] = _glorot_uniform(rng, (256, 128))
    P['prenet/dense_2/bias'] = np.zeros(128)
    _cbhg(P, rng, 'encoder_cbhg', 16, 128, [128, 128])                    # modules.py:15-22
    P['attention/memory_layer/kernel'] = _glorot_uniform(rng, (256, 256))  # tacotron.py:68
    P['attention/query_layer/kernel'] = _glorot_uniform(rng, (256, 256))
    P['attention/attention_v'] = _glorot_uniform(rng, (256,))
    P['decoder_prenet/dense_1/kernel'] = _glorot_uniform(rng, (num_mels + 256, 256))  # rnn_wrappers.py:23
    P['decoder_prenet/dense_1/bias'] = np.zeros(256)
    P['decoder_prenet/dense_2/kernel'] = _glorot_uniform(rng, (256, 128))
    P['decoder_prenet/dense_2/bias'] = np.zeros(128)
    _gru(P, rng, 'attention_gru', 128, 256)                                # tacotron.py:67
    P['concat_projection/kernel'] = _glorot_uniform(rng, (512, 256))       # tacotron.py:77
    P['concat_projection/bias'] = np.zeros(256)
    _gru(P, rng, 'decoder_gru_1', 256, 256)                                # tacotron.py:78
    _gru(P, rng, 'decoder_gru_2', 256, 256)                                # tacotron.py:79
    P['output_projection/kernel'] = _glorot_uniform(rng, (256, num_mels * r))  # tacotron.py:83
    P['output_projection/bias'] = np.zeros(num_mels * r)
    _cbhg(P, rng, 'post_cbhg', 8, num_mels, [256, num_mels])               # modules.py:25-32
    P['linear/kernel'] = _glorot_uniform(rng, (256, num_freq))             # tacotron.py:101
    P['linear/bias'] = np.zeros(num_freq)
    return P


# --------------------------------------------------------------------------------------
# Operators
# --------------------------------------------------------------------------------------
def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def dense(x, W, b=None):
    """A.1 tf.layers.dense over the last axis."""
    y = x @ W
    return y if b is None else y + b


def conv1d_same(x, W, b):
    """A.3 tf.layers.conv1d(padding='same', stride 1): cross-correlation, kernel [k,Cin,Cout],
    pad_left=(k-1)//2, pad_right=k-1-pad_left, zeros."""
    k = W.shape[0]
    pl = (k - 1) // 2
    pr = k - 1 - pl
    T = x.shape[1]
    xp = np.pad(x, ((0, 0), (pl, pr), (0, 0)))
    y = np.zeros(x.shape[:2] + (W.shape[2],), dtype=x.dtype) + b
    for j in range(k):
        y += xp[:, j:j + T, :] @ W[j]
    return y


def batch_norm_train(x, gamma, beta):
    """A.4 training-mode batch norm over (N,T) incl. padded frames, biased variance."""
    mu = x.mean(axis=(0, 1))
    var = x.var(axis=(0, 1))
    return (x - mu) / np.sqrt(var + BN_EPS) * gamma + beta, mu, var


def batch_norm_infer(x, gamma, beta, mm, mv):
    return (x - mm) / np.sqrt(mv + BN_EPS) * gamma + beta


def conv1d_bn(x, P, scope, act, training, stats):
    """modules.py:93-101: conv -> activation -> batch norm (in that order)."""
    y = conv1d_same(x, P[scope + '/kernel'], P[scope + '/bias'])
    if act == 'relu':
        y = np.maximum(y, 0.0)
    if training:
        y, mu, var = batch_norm_train(y, P[scope + '/gamma'], P[scope + '/beta'])
        stats[scope] = (mu, var)
        return y
    return batch_norm_infer(y, P[scope + '/gamma'], P[scope + '/beta'],
                            P[scope + '/moving_mean'], P[scope + '/moving_variance'])


def maxpool2_same(x):
    """modules.py:45-49 max_pooling1d(pool 2, stride 1, 'same') = max(x[t], x[t+1]); the last
    frame sees only itself (right pad is -inf)."""
    y = x.copy()
    y[:, :-1, :] = np.maximum(x[:, :-1, :], x[:, 1:, :])
    return y


def highway(x, P, scope):
    """modules.py:77-90."""
    H = np.maximum(dense(x, P[scope + '/H/kernel'], P[scope + '/H/bias']), 0.0)
    T = sigmoid(dense(x, P[scope + '/T/kernel'], P[scope + '/T/bias']))
    return H * T + x * (1.0 - T)


def gru_cell(x, h, P, scope):
    """A.5 tf.contrib.rnn.GRUCell: [r,u]=sigmoid([x,h]Wg+bg); c=tanh([x,r*h]Wc+bc);
    h'=u*h+(1-u)*c.  Gate order r then u, reset applied before the candidate matmul."""
    n = h.shape[-1]
    g = sigmoid(np.concatenate([x, h], -1) @ P[scope + '/gates/kernel'] + P[scope + '/gates/bias'])
    r, u = g[..., :n], g[..., n:]
    c = np.tanh(np.concatenate([x, r * h], -1) @ P[scope + '/candidate/kernel'] + P[scope + '/candidate/bias'])
    return u * h + (1.0 - u) * c


def bigru(x, lengths, P, scope):
    """A.6 tf.nn.bidirectional_dynamic_rnn (modules.py:68-74).  lengths=None -> full length.
    Output rows at t >= length are zero; the state is carried unchanged past the length."""
    N, T, _ = x.shape
    n = 128
    if lengths is None:
        lengths = np.full(N, T, dtype=np.int64)
    out_f = np.zeros((N, T, n), dtype=x.dtype)
    out_b = np.zeros((N, T, n), dtype=x.dtype)
    h = np.zeros((N, n), dtype=x.dtype)
    for t in range(T):
        hn = gru_cell(x[:, t, :], h, P, scope + '/gru_fw')
        m = (t < lengths)[:, None]
        h = np.where(m, hn, h)
        out_f[:, t, :] = np.where(m, hn, 0.0)
    # backward: reverse_sequence by length == run t = L-1 .. 0 from a zero state per row
    h = np.zeros((N, n), dtype=x.dtype)
    for t in range(T - 1, -1, -1):
        hn = gru_cell(x[:, t, :], h, P, scope + '/gru_bw')
        m = (t < lengths)[:, None]
        h = np.where(m, hn, h)
        out_b[:, t, :] = np.where(m, hn, 0.0)
    return np.concatenate([out_f, out_b], axis=2)


def cbhg(x, lengths, P, scope, K, training, stats):
    """modules.py:35-74."""
    bank = np.concatenate(
        [conv1d_bn(x, P, '%s/conv_bank/conv1d_%d' % (scope, k), 'relu', training, stats)
         for k in range(1, K + 1)], axis=-1)
    pooled = maxpool2_same(bank)
    p1 = conv1d_bn(pooled, P, scope + '/proj_1', 'relu', training, stats)
    p2 = conv1d_bn(p1, P, scope + '/proj_2', None, training, stats)
    hw = p2 + x
    if hw.shape[2] != 128:
        hw = dense(hw, P[scope + '/highway_dense/kernel'], P[scope + '/highway_dense/bias'])
    for i in range(1, 5):
        hw = highway(hw, P, '%s/highway_%d' % (scope, i))
    return bigru(hw, lengths, P, scope)


def prenet(x, P, scope):
    """modules.py:5-12; dropout is inert in this fork (training= never passed)."""
    x = np.maximum(dense(x, P[scope + '/dense_1/kernel'], P[scope + '/dense_1/bias']), 0.0)
    x = np.maximum(dense(x, P[scope + '/dense_2/kernel'], P[scope + '/dense_2/bias']), 0.0)
    return x


def softmax(x, axis=-1):
    m = x.max(axis=axis, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis=axis, keepdims=True)


def decoder_train(enc_out, mel_targets, P, r, num_mels=80):
    """tacotron.py:66-97 + rnn_wrappers.py + helpers.py:41-82 (teacher forcing), Appendix A.7/A.8.
    Returns decoder_outputs [N,S,num_mels*r] and alignments [N,Ti,S]."""
    N, Ti, _ = enc_out.shape
    To = mel_targets.shape[1]
    assert To % r == 0
    S = To // r
    keys = enc_out @ P['attention/memory_layer/kernel']            # A.7 keys = memory . W_mem
    v = P['attention/attention_v']
    h_att = np.zeros((N, 256), dtype=enc_out.dtype)                # zero_state (tacotron.py:84)
    ctx = np.zeros((N, 256), dtype=enc_out.dtype)
    g1 = np.zeros((N, 256), dtype=enc_out.dtype)
    g2 = np.zeros((N, 256), dtype=enc_out.dtype)
    outs = np.zeros((N, S, num_mels * r), dtype=enc_out.dtype)
    aligns = np.zeros((N, Ti, S), dtype=enc_out.dtype)
    for s in range(S):
        # helpers.py:49,76: go frame zeros at s=0, then every r-th target frame
        frame = np.zeros((N, num_mels), dtype=enc_out.dtype) if s == 0 else mel_targets[:, r * s - 1, :]
        cell_in = np.concatenate([frame, ctx], -1)                 # AttentionWrapper: [inputs, attention]
        p = prenet(cell_in, P, 'decoder_prenet')                   # rnn_wrappers.py:22-24
        h_att = gru_cell(p, h_att, P, 'attention_gru')
        q = h_att @ P['attention/query_layer/kernel']
        score = np.einsum('ntd,d->nt', np.tanh(keys + q[:, None, :]), v)
        a = softmax(score, axis=1)                                 # over ALL Ti (no memory mask)
        ctx = np.einsum('nt,ntd->nd', a, enc_out)
        aligns[:, :, s] = a
        y = np.concatenate([h_att, ctx], -1) @ P['concat_projection/kernel'] + P['concat_projection/bias']
        g1 = gru_cell(y, g1, P, 'decoder_gru_1')
        d1 = y + g1                                                # ResidualWrapper
        g2 = gru_cell(d1, g2, P, 'decoder_gru_2')
        d2 = d1 + g2
        outs[:, s, :] = d2 @ P['output_projection/kernel'] + P['output_projection/bias']
    return outs, aligns


def decoder_infer(enc_out, P, r, max_iters, num_mels=80):
    """Free-running decode: models/helpers.py:7-38 (TacoTestHelper feeds the LAST of the r predicted frames back, stops
    when a whole r-frame output is exactly zero -- checked after the step -- or at max_iters), tacotron.py:86-94."""
    N, Ti, _ = enc_out.shape
    keys = enc_out @ P['attention/memory_layer/kernel']
    v = P['attention/attention_v']
    h_att = np.zeros((N, 256)); ctx = np.zeros((N, 256)); g1 = np.zeros((N, 256)); g2 = np.zeros((N, 256))
    frame = np.zeros((N, num_mels))
    outs, aligns = [], []
    finished = np.zeros(N, dtype=bool)
    for s in range(max_iters):
        p = prenet(np.concatenate([frame, ctx], -1), P, 'decoder_prenet')
        h_att = gru_cell(p, h_att, P, 'attention_gru')
        q = h_att @ P['attention/query_layer/kernel']
        score = np.einsum('ntd,d->nt', np.tanh(keys + q[:, None, :]), v)
        a = softmax(score, axis=1)
        ctx = np.einsum('nt,ntd->nd', a, enc_out)
        y = np.concatenate([h_att, ctx], -1) @ P['concat_projection/kernel'] + P['concat_projection/bias']
        g1 = gru_cell(y, g1, P, 'decoder_gru_1'); d1 = y + g1
        g2 = gru_cell(d1, g2, P, 'decoder_gru_2'); d2 = d1 + g2
        o = d2 @ P['output_projection/kernel'] + P['output_projection/bias']
        outs.append(o); aligns.append(a)
        finished |= np.all(o == 0.0, axis=1)
        if finished.all():
            break
        frame = o[:, -num_mels:]
    return np.stack(outs, 1), np.stack(aligns, 2)


def forward_infer(P, inputs, input_lengths, identities=None, id_num=0, r=5, max_iters=2000, num_mels=80):
    """models/tacotron.py:35-104 with linear_targets=None: batch norm uses the moving statistics."""
    emb = P['embedding'][inputs]
    if identities is not None and id_num > 1:
        eid = P['embedding_id'][identities][:, None, :]
        emb = np.concatenate([emb, np.tile(eid, (1, inputs.shape[1], 1))], axis=2)
    pre = prenet(emb, P, 'prenet')
    enc = cbhg(pre, input_lengths, P, 'encoder_cbhg', 16, False, None)
    dec, aligns = decoder_infer(enc, P, r, max_iters, num_mels)
    mel_out = dec.reshape(inputs.shape[0], -1, num_mels)
    post = cbhg(mel_out, None, P, 'post_cbhg', 8, False, None)
    lin_out = dense(post, P['linear/kernel'], P['linear/bias'])
    return dict(mel_outputs=mel_out, linear_outputs=lin_out, alignments=aligns, encoder_outputs=enc)


def forward(P, inputs, input_lengths, mel_targets, identities=None, id_num=0, r=5,
            num_mels=80, training=True):
    """models/tacotron.py:35-104.  Returns dict(mel_outputs, linear_outputs, alignments, bn_stats)."""
    stats = OrderedDict()
    emb = P['embedding'][inputs]                                           # tacotron.py:46
    if identities is not None and id_num > 1:                              # tacotron.py:48-55
        eid = P['embedding_id'][identities][:, None, :]
        emb = np.concatenate([emb, np.tile(eid, (1, inputs.shape[1], 1))], axis=2)
    pre = prenet(emb, P, 'prenet')                                         # tacotron.py:62
    enc = cbhg(pre, input_lengths, P, 'encoder_cbhg', 16, training, stats)  # tacotron.py:63
    dec, aligns = decoder_train(enc, mel_targets, P, r, num_mels)
    N = inputs.shape[0]
    mel_out = dec.reshape(N, -1, num_mels)                                 # tacotron.py:97
    post = cbhg(mel_out, None, P, 'post_cbhg', 8, training, stats)         # tacotron.py:100
    lin_out = dense(post, P['linear/kernel'], P['linear/bias'])            # tacotron.py:101
    return dict(mel_outputs=mel_out, linear_outputs=lin_out, alignments=aligns, bn_stats=stats,
                encoder_outputs=enc, prenet_outputs=pre, post_outputs=post)


def n_priority_freq(sample_rate=20000, num_freq=1025):
    """tacotron.py:134."""
    return int(3000 / (sample_rate * 0.5) * num_freq)


def loss(mel_out, lin_out, mel_targets, linear_targets, sample_rate=20000):
    """tacotron.py:127-137 (the optional regularisers :140-171 are alignment_regularity below)."""
    mel_loss = np.abs(mel_targets - mel_out).mean()
    l1 = np.abs(linear_targets - lin_out)
    npf = n_priority_freq(sample_rate, lin_out.shape[-1])
    linear_loss = 0.5 * l1.mean() + 0.5 * l1[:, :, :npf].mean()
    return mel_loss + linear_loss, mel_loss, linear_loss


def alignment_regularity(alignments, overwrought=0.0, oneorder_dynamic=0.0, variance_between_row=0.0,
                         alignment_entropy=0.0):
    """loss_regularity of tacotron.py:140-171 (all weights default to 0.0 in hparams.py:74-77 of the reference).
    alignments: [N, T_in, S] as returned by the model (tacotron.py:104).  NB the reference applies a SECOND softmax over
    the decoder-step axis (:142) to the already normalised alignments; pr_slice_last_word (:158) is computed but unused."""
    a = np.asarray(alignments, np.float64)
    N, Ti, S = a.shape
    reg = 0.0
    if overwrought or oneorder_dynamic or variance_between_row or alignment_entropy:
        e = np.exp(a - a.max(axis=2, keepdims=True))
        pr = e / e.sum(axis=2, keepdims=True)
        if alignment_entropy:
            reg += -1.0 * alignment_entropy * np.mean(pr * np.log(pr))
        if oneorder_dynamic:
            reg += oneorder_dynamic * np.abs(pr[:, :, :-1] - pr[:, :, 1:]).sum()
        if overwrought:
            size = S - 41                                   # tf.slice(pr, [0,0,40], [N,1,S-41]); -1 = "to the end"
            if size < -1 or S < 40:
                raise ValueError('overwrought regulariser needs at least 40 decoder steps (tf.slice at tacotron.py:159)')
            end = S if size == -1 else 40 + size
            reg += overwrought * pr[:, 0:1, 40:end].sum()
        if variance_between_row:
            sum_row = a.sum(axis=2)
            mean_row = sum_row.mean(axis=1, keepdims=True)
            reg += variance_between_row * ((mean_row - sum_row) ** 2).sum()
    return reg


def noam_lr(init_lr, global_step):
    """tacotron.py:198-202."""
    warm = 4000.0
    step = float(global_step + 1)
    return init_lr * warm ** 0.5 * min(step * warm ** -1.5, step ** -0.5)


def adam_step(P, G, M, V, t, lr, beta1=0.9, beta2=0.999, eps=1e-8, clip=1.0, sparse_sumsq=None):
    """tf.clip_by_global_norm(...,1.0) + tf.train.AdamOptimizer (A.10, A.11).
    t = 1-based apply count.  sparse_sumsq: {name: sum of squares of the UN-deduplicated
    IndexedSlices rows} for the embedding tables (A.11 quirk); None -> dense norm.
    Updates P, M, V in place; returns the global norm."""
    sq = 0.0
    for k, g in G.items():
        if sparse_sumsq is not None and k in sparse_sumsq:
            sq += sparse_sumsq[k]
        else:
            sq += float((g * g).sum())
    norm = np.sqrt(sq)
    scale = 1.0 / max(norm, clip) * clip
    lr_t = lr * np.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    for k, g in G.items():
        g = g * scale
        M[k] = beta1 * M[k] + (1.0 - beta1) * g
        V[k] = beta2 * V[k] + (1.0 - beta2) * g * g
        P[k] = P[k] - lr_t * M[k] / (np.sqrt(V[k]) + eps)
    return norm


def bn_moving_update(P, stats):
    """UPDATE_OPS (tacotron.py:193): mov -= (mov - batch) * (1 - momentum); biased batch variance."""
    for scope, (mu, var) in stats.items():
        P[scope + '/moving_mean'] = P[scope + '/moving_mean'] - (P[scope + '/moving_mean'] - mu) * (1.0 - BN_MOMENTUM)
        P[scope + '/moving_variance'] = P[scope + '/moving_variance'] - (P[scope + '/moving_variance'] - var) * (1.0 - BN_MOMENTUM)


# --------------------------------------------------------------------------------------
# Synthetic LJSpeech-shaped batch (SURVEY.md 8(d) "Synthetic inputs")
# --------------------------------------------------------------------------------------
def synth_batch(N, Ti, To, r, seed=1234, id_num=0, num_mels=80, num_freq=1025, vocab=NUM_SYMBOLS2):
    assert To % r == 0
    rng = np.random.RandomState(seed)
    lens = rng.randint(int(np.ceil(0.6 * Ti)), Ti + 1, size=N)
    lens[rng.randint(N)] = Ti
    inputs = np.zeros((N, Ti), dtype=np.int32)
    for n in range(N):
        inputs[n, :lens[n] - 1] = rng.randint(2, vocab, size=lens[n] - 1)
        inputs[n, lens[n] - 1] = 1                       # EOS (text/__init__.py:61)
    nfr = rng.randint(int(np.ceil(0.7 * To)), To, size=N)  # <= To-1 (feeder leaves >=1 pad frame)
    nfr[rng.randint(N)] = To - 1
    mel = rng.uniform(0, 1, size=(N, To, num_mels)).astype(np.float32)
    lin = rng.uniform(0, 1, size=(N, To, num_freq)).astype(np.float32)
    for n in range(N):
        mel[n, nfr[n]:] = 0.0
        lin[n, nfr[n]:] = 0.0
    ids = rng.randint(0, id_num, size=N).astype(np.int32) if id_num > 1 else None
    return dict(inputs=inputs, input_lengths=lens.astype(np.int32), mel_targets=mel,
                linear_targets=lin, identities=ids)
