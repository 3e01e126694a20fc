"""ORACLE (test infrastructure, NOT product code) -- independent PyTorch-CPU composition of the
same training step as oracle/tacotron_np.py (forward, loss, autograd backward, global-norm clip,
TF-style Adam, BN moving stats).  Two uses only:
  * float64: gradient / post-step oracle for tests (cross-checked against tacotron_np forward);
  * float32: the "CPU stand-in" baseline timed by bench.py's cpu_baseline leg (kind "port":
    TF-1 cannot run anywhere in this pipeline).

PARITY UNPINNED by the reference (no TF, no reference tests) -- see tacotron_np.py header.
Built from torch library ops (F.conv1d(padding='same'), F.linear, F.batch_norm; the max-pool is an explicit
first-max autograd function, cross-checked against F.max_pool1d) rather than the explicit loops of tacotron_np
so that the two restatements are independent.
Reference call sites are cited per function.
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3
BN_MOMENTUM = 0.99
DEBUG_TAPS = None          # set to a dict to capture CBHG intermediates (developer diagnostics only)

# ---- piecewise-linear decisions (ReLU on/off, max-pool argmax) ---------------------------------------------------------
# The training step is piecewise smooth: a ReLU whose pre-activation, or a pooling window whose two operands, agree with the
# kink to within fp32 rounding can legitimately fall on the other side in an fp32 implementation, and ONE such unit moves a
# small gradient tensor by a few 1e-3 of its norm.  To compare gradients on the SAME linear piece the oracle can
#   record  its own decisions and their margins (distance from the kink), and
#   replay  decisions supplied by the caller (the HIP path's, read back from its saved activations).
# tests/decisions.py asserts that every replayed decision that differs from the recorded one has a margin within fp32
# rounding of the kink, i.e. the HIP path never takes a decision the float64 oracle would call clear.
# DECISIONS = None (default: plain ops) | dict(mode='record'|'replay', masks={site: bool tensor}, margins={site: tensor}).
# Sites (channel-last shapes): 'prenet/dense_{1,2}' [N,Ti,*]; '<cbhg>/conv_bank/conv1d_k', '<cbhg>/proj_1' [N,T,C];
# '<cbhg>/pool' [N,T,K*128] (True = first operand wins); '<cbhg>/highway_i/H' [N,T,128]; 'decoder_prenet/dense_{1,2}@s' [N,*].
DECISIONS = None


def relu_site(x, site, channel_first=False):
    d = DECISIONS
    if d is None:
        return F.relu(x)
    cl = (lambda t: t.transpose(1, 2)) if channel_first else (lambda t: t)
    if d['mode'] == 'record':
        d['masks'][site] = cl((x > 0).detach())
        d['margins'][site] = cl(x.detach().abs())
        return F.relu(x)
    return x * cl(d['masks'][site]).to(x.dtype)          # value and gradient follow the supplied on/off decision



def to_torch(P_np, dtype=torch.float64, requires_grad=True):
    P = OrderedDict()
    for k, v in P_np.items():
        t = torch.tensor(np.asarray(v), dtype=dtype)
        if requires_grad and not k.endswith(('/moving_mean', '/moving_variance')):
            t.requires_grad_(True)
        P[k] = t
    return P


def conv1d_bn(x, P, scope, act, training, stats):
    """modules.py:93-101 (conv -> act -> BN).  x: [N,T,C] channel-last like the reference."""
    W = P[scope + '/kernel']                       # [k,Cin,Cout] (TF layout)
    y = F.conv1d(x.transpose(1, 2), W.permute(2, 1, 0), P[scope + '/bias'], padding='same')
    if act == 'relu':
        y = relu_site(y, scope, channel_first=True)
    if training:
        mu = y.mean(dim=(0, 2))
        var = y.var(dim=(0, 2), unbiased=False)
        stats[scope] = (mu.detach(), var.detach())
        y = F.batch_norm(y, None, None, P[scope + '/gamma'], P[scope + '/beta'], True, 0.0, BN_EPS)
    else:
        y = F.batch_norm(y, P[scope + '/moving_mean'], P[scope + '/moving_variance'],
                         P[scope + '/gamma'], P[scope + '/beta'], False, 0.0, BN_EPS)
    return y.transpose(1, 2)


class _MaxPool2SameFirstMax(torch.autograd.Function):
    """y[t] = x[t] where first[t] else x[t+1] (first[T-1] is always True): the max-pool of modules.py:45-49 with an explicit
    argmax; the gradient follows the same routing."""

    @staticmethod
    def forward(ctx, x, first):
        nxt = torch.cat([x[:, 1:], x[:, -1:]], dim=1)
        ctx.save_for_backward(first)
        return torch.where(first, x, nxt)

    @staticmethod
    def backward(ctx, g):
        first, = ctx.saved_tensors
        g1 = torch.where(first, g, torch.zeros_like(g))
        g2 = g - g1                                    # gradient of the outputs whose maximum is x[t+1]
        dx = g1.clone()
        dx[:, 1:] += g2[:, :-1]
        return dx, None


TIE_REL = {torch.float64: 1e-11, torch.float32: 0.0}      # fp32 (CPU baseline timing only): plain bitwise first max


def maxpool2_same(x, site=None):
    """modules.py:45-49: pool 2, stride 1, 'same' (right pad -inf, so the last frame passes through).  The gradient goes to
    the FIRST maximum (TF CPU MaxPoolGrad and the HIP kernel).  In float64 two operands closer than 1e-11 of the largest
    magnitude count as tied: windows that are equal in exact arithmetic (the conv bank over padded text, where every
    position carries embedding[0]) leave a library conv / batch norm differing in the last bits, and a bitwise argmax would
    route their gradient by rounding noise.  Equals F.max_pool1d on tie-free inputs (tests/test_oracle.py)."""
    d = DECISIONS
    if d is not None and d['mode'] == 'replay' and site is not None:
        first = d['masks'][site].clone()
        first[:, -1] = True
        return _MaxPool2SameFirstMax.apply(x, first)
    xd = x.detach()
    nxt = torch.cat([xd[:, 1:], torch.full_like(xd[:, :1], float('-inf'))], dim=1)
    tol = TIE_REL.get(x.dtype, 0.0) * float(xd.abs().max()) if x.numel() else 0.0
    first = xd >= nxt - tol
    if d is not None and site is not None:
        d['masks'][site] = first
        d['margins'][site] = (xd - nxt).abs()
    return _MaxPool2SameFirstMax.apply(x, first)


def maxpool2_same_library(x):
    """library form (F.max_pool1d), kept as the independent cross-check of maxpool2_same on tie-free inputs."""
    xt = F.pad(x.transpose(1, 2), (0, 1), value=float('-inf'))
    return F.max_pool1d(xt, 2, 1).transpose(1, 2)


def gru_cell(x, h, P, scope):
    """Appendix A.5 (tf.contrib.rnn.GRUCell)."""
    n = h.shape[-1]
    g = torch.sigmoid(F.linear(torch.cat([x, h], -1), P[scope + '/gates/kernel'].t(), P[scope + '/gates/bias']))
    r, u = g[..., :n], g[..., n:]
    c = torch.tanh(F.linear(torch.cat([x, r * h], -1), P[scope + '/candidate/kernel'].t(), P[scope + '/candidate/bias']))
    return u * h + (1.0 - u) * c


def bigru(x, lengths, P, scope):
    """Appendix A.6 (modules.py:68-74)."""
    N, T, _ = x.shape
    n = 128
    if lengths is None:
        lengths = torch.full((N,), T, dtype=torch.long)
    lengths = torch.as_tensor(lengths, dtype=torch.long)
    zeros = torch.zeros(N, n, dtype=x.dtype)
    outs_f, outs_b = [None] * T, [None] * T
    h = zeros
    for t in range(T):
        hn = gru_cell(x[:, t], h, P, scope + '/gru_fw')
        m = (t < lengths)[:, None]
        h = torch.where(m, hn, h)
        outs_f[t] = torch.where(m, hn, zeros)
    h = zeros
    for t in range(T - 1, -1, -1):
        hn = gru_cell(x[:, t], h, P, scope + '/gru_bw')
        m = (t < lengths)[:, None]
        h = torch.where(m, hn, h)
        outs_b[t] = torch.where(m, hn, zeros)
    return torch.cat([torch.stack(outs_f, 1), torch.stack(outs_b, 1)], dim=2)


def highway(x, P, scope):
    H = relu_site(F.linear(x, P[scope + '/H/kernel'].t(), P[scope + '/H/bias']), scope + '/H')
    T = torch.sigmoid(F.linear(x, P[scope + '/T/kernel'].t(), P[scope + '/T/bias']))
    return H * T + x * (1.0 - T)


def cbhg(x, lengths, P, scope, K, training, stats):
    """modules.py:35-74."""
    bank = torch.cat([conv1d_bn(x, P, '%s/conv_bank/conv1d_%d' % (scope, k), 'relu', training, stats)
                      for k in range(1, K + 1)], dim=-1)
    pooled = maxpool2_same(bank, scope + '/pool')
    if DEBUG_TAPS is not None:                       # scripts/dev_tie_debug.py: keep intermediates (and their gradients)
        for nm, tns in (('bank', bank), ('pooled', pooled)):
            if tns.requires_grad:
                tns.retain_grad()
            DEBUG_TAPS[scope + '/' + nm] = tns
    p1 = conv1d_bn(pooled, P, scope + '/proj_1', 'relu', training, stats)
    p2 = conv1d_bn(p1, P, scope + '/proj_2', None, training, stats)
    hw = p2 + x
    if hw.shape[2] != 128:
        hw = F.linear(hw, P[scope + '/highway_dense/kernel'].t(), P[scope + '/highway_dense/bias'])
    for i in range(1, 5):
        hw = highway(hw, P, '%s/highway_%d' % (scope, i))
    return bigru(hw, lengths, P, scope)


def prenet(x, P, scope, step=None):
    tag = '' if step is None else '@%d' % step
    x = relu_site(F.linear(x, P[scope + '/dense_1/kernel'].t(), P[scope + '/dense_1/bias']), scope + '/dense_1' + tag)
    return relu_site(F.linear(x, P[scope + '/dense_2/kernel'].t(), P[scope + '/dense_2/bias']), scope + '/dense_2' + tag)


def decoder_train(enc, mel_targets, P, r, num_mels=80):
    """tacotron.py:66-97, rnn_wrappers.py:22-24,50-52, helpers.py:41-82; Appendix A.7/A.8."""
    N, Ti, _ = enc.shape
    S = mel_targets.shape[1] // r
    keys = enc @ P['attention/memory_layer/kernel']
    v = P['attention/attention_v']
    z = torch.zeros(N, 256, dtype=enc.dtype)
    h_att, ctx, g1, g2 = z, z, z, z
    outs, aligns = [], []
    for s in range(S):
        frame = torch.zeros(N, num_mels, dtype=enc.dtype) if s == 0 else mel_targets[:, r * s - 1, :]
        p = prenet(torch.cat([frame, ctx], -1), P, 'decoder_prenet', step=s)
        h_att = gru_cell(p, h_att, P, 'attention_gru')
        q = h_att @ P['attention/query_layer/kernel']
        score = (torch.tanh(keys + q[:, None, :]) * v).sum(-1)
        a = torch.softmax(score, dim=1)
        ctx = torch.bmm(a[:, None, :], enc)[:, 0, :]
        aligns.append(a)
        y = F.linear(torch.cat([h_att, ctx], -1), P['concat_projection/kernel'].t(), P['concat_projection/bias'])
        g1 = gru_cell(y, g1, P, 'decoder_gru_1')
        d1 = y + g1
        g2 = gru_cell(d1, g2, P, 'decoder_gru_2')
        d2 = d1 + g2
        outs.append(F.linear(d2, P['output_projection/kernel'].t(), P['output_projection/bias']))
    return torch.stack(outs, 1), torch.stack(aligns, 2)


def forward(P, inputs, input_lengths, mel_targets, identities=None, id_num=0, r=5, num_mels=80,
            training=True):
    """models/tacotron.py:35-104."""
    stats = OrderedDict()
    inputs = torch.as_tensor(np.asarray(inputs), dtype=torch.long)
    emb = F.embedding(inputs, P['embedding'])
    if identities is not None and id_num > 1:
        ids = torch.as_tensor(np.asarray(identities), dtype=torch.long)
        eid = F.embedding(ids, P['embedding_id'])[:, None, :].expand(-1, inputs.shape[1], -1)
        emb = torch.cat([emb, eid], dim=2)
    emb.retain_grad() if emb.requires_grad else None
    pre = prenet(emb, P, 'prenet')
    enc = cbhg(pre, input_lengths, P, 'encoder_cbhg', 16, training, stats)
    dec, aligns = decoder_train(enc, mel_targets, P, r, num_mels)
    mel_out = dec.reshape(inputs.shape[0], -1, num_mels)
    post = cbhg(mel_out, None, P, 'post_cbhg', 8, training, stats)
    lin_out = F.linear(post, P['linear/kernel'].t(), P['linear/bias'])
    return dict(mel_outputs=mel_out, linear_outputs=lin_out, alignments=aligns, bn_stats=stats,
                encoder_outputs=enc, embedded_inputs=emb, post_outputs=post)


def loss_fn(mel_out, lin_out, mel_targets, linear_targets, sample_rate=20000):
    """tacotron.py:127-137."""
    mel_loss = (mel_targets - mel_out).abs().mean()
    l1 = (linear_targets - lin_out).abs()
    npf = int(3000 / (sample_rate * 0.5) * lin_out.shape[-1])
    linear_loss = 0.5 * l1.mean() + 0.5 * l1[:, :, :npf].mean()
    return mel_loss + linear_loss, mel_loss, linear_loss


def alignment_regularity(alignments, overwrought=0.0, oneorder_dynamic=0.0, variance_between_row=0.0,
                         alignment_entropy=0.0):
    """tacotron.py:140-171 on alignments [N, T_in, S] (second softmax over the decoder-step axis, :142)."""
    a = alignments
    S = a.shape[2]
    reg = a.new_zeros(())
    if overwrought or oneorder_dynamic or variance_between_row or alignment_entropy:
        pr = torch.softmax(a, dim=2)
        if alignment_entropy:
            reg = reg - alignment_entropy * (pr * pr.log()).mean()
        if oneorder_dynamic:
            reg = reg + oneorder_dynamic * (pr[:, :, :-1] - pr[:, :, 1:]).abs().sum()
        if overwrought:
            size = S - 41
            if size < -1 or S < 40:
                raise ValueError('overwrought regulariser needs at least 40 decoder steps (tf.slice at tacotron.py:159)')
            end = S if size == -1 else 40 + size
            reg = reg + overwrought * pr[:, 0:1, 40:end].sum()
        if variance_between_row:
            sum_row = a.sum(dim=2)
            mean_row = sum_row.mean(dim=1, keepdim=True)
            reg = reg + variance_between_row * ((mean_row - sum_row) ** 2).sum()
    return reg


def noam_lr(init_lr, global_step):
    """tacotron.py:198-202."""
    step = float(global_step + 1)
    return init_lr * 4000.0 ** 0.5 * min(step * 4000.0 ** -1.5, step ** -0.5)


class TrainState:
    """Parameters + Adam slots + global_step; one .step(batch) == one sess.run of train.py:142-146."""

    def __init__(self, P_np, dtype=torch.float64, id_num=0, r=5, init_lr=0.002, decay=True,
                 beta1=0.9, beta2=0.999, tf_sparse_norm=True, sample_rate=20000, regularity=None):
        self.P = to_torch(P_np, dtype)
        self.dtype = dtype
        self.id_num, self.r = id_num, r
        self.init_lr, self.decay, self.beta1, self.beta2 = init_lr, decay, beta1, beta2
        self.tf_sparse_norm = tf_sparse_norm
        self.sample_rate = sample_rate
        self.regularity = dict(regularity or {})      # overwrought / oneorder_dynamic / variance_between_row / alignment_entropy
        self.global_step = 0
        self.M = {k: torch.zeros_like(v) for k, v in self.P.items() if v.requires_grad}
        self.V = {k: torch.zeros_like(v) for k, v in self.P.items() if v.requires_grad}
        self.last = None

    def forward_backward(self, batch):
        P = self.P
        for v in P.values():
            v.grad = None
        mel_t = torch.as_tensor(batch['mel_targets'], dtype=self.dtype)
        lin_t = torch.as_tensor(batch['linear_targets'], dtype=self.dtype)
        out = forward(P, batch['inputs'], batch['input_lengths'], mel_t, batch.get('identities'),
                      self.id_num, self.r, mel_t.shape[-1], training=True)
        loss, mel_loss, lin_loss = loss_fn(out['mel_outputs'], out['linear_outputs'], mel_t, lin_t,
                                           self.sample_rate)
        reg = alignment_regularity(out['alignments'], **self.regularity)
        loss = loss + reg
        loss.backward()
        grads = OrderedDict((k, v.grad if v.grad is not None else torch.zeros_like(v))
                            for k, v in P.items() if v.requires_grad)
        # A.11: tf.global_norm squares the un-deduplicated IndexedSlices rows of the two embeddings
        sparse = {}
        eg = out['embedded_inputs'].grad
        if eg is not None:
            et = P['embedding'].shape[1]
            sparse['embedding'] = float((eg[:, :, :et] ** 2).sum())
            if 'embedding_id' in P and eg.shape[2] > et:
                sparse['embedding_id'] = float((eg[:, :, et:].sum(dim=1) ** 2).sum())
        self.last = dict(out=out, loss=float(loss.detach()), loss_regularity=float(reg.detach()), mel_loss=float(mel_loss.detach()), linear_loss=float(lin_loss.detach()),
                         grads=grads, sparse_sumsq=sparse)
        return self.last

    def apply(self, last=None, clip=1.0, eps=1e-8):
        last = last or self.last
        grads = last['grads']
        sq = 0.0
        for k, g in grads.items():
            if self.tf_sparse_norm and k in last['sparse_sumsq']:
                sq += last['sparse_sumsq'][k]
            else:
                sq += float((g * g).sum())
        norm = math.sqrt(sq)
        scale = clip / max(norm, clip)
        lr = noam_lr(self.init_lr, self.global_step) if self.decay else self.init_lr
        t = self.global_step + 1
        lr_t = lr * math.sqrt(1.0 - self.beta2 ** t) / (1.0 - self.beta1 ** t)
        with torch.no_grad():
            for k, g in grads.items():
                g = g * scale
                self.M[k].mul_(self.beta1).add_(g, alpha=1.0 - self.beta1)
                self.V[k].mul_(self.beta2).addcmul_(g, g, value=1.0 - self.beta2)
                self.P[k].sub_(lr_t * self.M[k] / (self.V[k].sqrt() + eps))
            for scope, (mu, var) in last['out']['bn_stats'].items():   # UPDATE_OPS, tacotron.py:193
                mm, mv = self.P[scope + '/moving_mean'], self.P[scope + '/moving_variance']
                mm.sub_((mm - mu) * (1.0 - BN_MOMENTUM))
                mv.sub_((mv - var) * (1.0 - BN_MOMENTUM))
        self.global_step += 1
        return dict(global_norm=norm, learning_rate=lr)

    def step(self, batch):
        last = self.forward_backward(batch)
        info = self.apply(last)
        return self.global_step, last['loss'], info
