"""Flat fp32 parameter store of the MI355X Tacotron engine.

All trainable variables of reference models/tacotron.py:35-104 live in ONE contiguous device buffer
(creation order of SURVEY.md Appendix B), so that the optimizer, the global-norm reduction and the RCCL
gradient all-reduce each touch a single allocation; gradient buckets in backward order are contiguous
suffix ranges.  The in-buffer layout is chosen for the kernels (GRU kernels split into input / recurrent
halves, highway H|T fused, conv bank packed, the 1025-wide linear layer padded to a 1028 leading dimension);
`load_named` / `export_named` convert from / to the TF-style per-variable tensors (shapes of
tf.get_variable in the reference: GRUCell gates kernel [in+H, 2H], candidate kernel [in+H, H], ...).
"""
from collections import OrderedDict

import numpy as np


def _pad4(n):
    return (n + 3) & ~3


class Entry:
    __slots__ = ('name', 'offset', 'shape', 'size')

    def __init__(self, name, offset, shape):
        self.name, self.offset, self.shape = name, offset, tuple(shape)
        self.size = int(np.prod(shape))


class ParamLayout:
    def __init__(self, vocab=7352, embed_text=256, embed_id=64, id_num=0, r=5, num_mels=80, num_freq=1025):
        self.vocab, self.Et, self.id_num, self.r = vocab, embed_text, id_num, r
        self.Es = embed_id if id_num > 1 else 0
        self.num_mels, self.num_freq = num_mels, num_freq
        self.ld_lin = _pad4(num_freq)
        self.entries = OrderedDict()
        self.bn_entries = OrderedDict()      # non-trainable moving statistics
        self._off = 0
        self._bnoff = 0
        E = self.Et + self.Es
        self._add('embedding', (vocab, self.Et))
        if self.Es:
            self._add('embedding_id', (id_num, self.Es))
        self.dense_start = self._off          # everything from here on has dense gradients
        self._dense('prenet/dense_1', E, 256)
        self._dense('prenet/dense_2', 256, 128)
        self._cbhg('encoder_cbhg', 16, 128, (128, 128))
        self._add('attention/memory_layer/kernel', (256, 256))
        self._add('attention/query_layer/kernel', (256, 256))
        self._add('attention/attention_v', (256,))
        self._dense('decoder_prenet/dense_1', num_mels + 256, 256)
        self._dense('decoder_prenet/dense_2', 256, 128)
        self._gru('attention_gru', 128, 256)
        self._dense('concat_projection', 512, 256)
        self._gru('decoder_gru_1', 256, 256)
        self._gru('decoder_gru_2', 256, 256)
        self._dense('output_projection', 256, num_mels * r)
        self._cbhg('post_cbhg', 8, num_mels, (256, num_mels))
        self._add('linear/kernel', (256, self.ld_lin))
        self._add('linear/bias', (self.ld_lin,))
        self.total = self._off
        self.bn_total = self._bnoff

    # ---- layout construction -------------------------------------------------------------------
    def _add(self, name, shape):
        e = Entry(name, self._off, shape)
        self.entries[name] = e
        self._off += _pad4(e.size)
        return e

    def _addbn(self, name, shape):
        e = Entry(name, self._bnoff, shape)
        self.bn_entries[name] = e
        self._bnoff += _pad4(e.size)

    def _dense(self, scope, cin, cout):
        self._add(scope + '/kernel', (cin, cout))
        self._add(scope + '/bias', (cout,))

    def _convbn(self, scope, kshape, cout):
        self._add(scope + '/kernel', kshape)
        for s in ('bias', 'gamma', 'beta'):
            self._add(scope + '/' + s, (cout,))
        self._addbn(scope + '/moving_mean', (cout,))
        self._addbn(scope + '/moving_variance', (cout,))

    def _gru(self, scope, n_in, n):
        self._add(scope + '/wx', (n_in, 3 * n))       # input half: gates (r|u) | candidate
        self._add(scope + '/bias', (3 * n,))
        self._add(scope + '/whg', (n, 2 * n))         # recurrent half of the gates kernel
        self._add(scope + '/whc', (n, n))             # recurrent half of the candidate kernel

    def _cbhg(self, scope, K, cin, proj):
        taps = K * (K + 1) // 2
        self._convbn(scope + '/conv_bank', (taps, cin, 128), K * 128)
        self._convbn(scope + '/proj_1', (3, K * 128, proj[0]), proj[0])
        self._convbn(scope + '/proj_2', (3, proj[0], proj[1]), proj[1])
        if proj[1] != 128:
            self._dense(scope + '/highway_dense', proj[1], 128)
        for i in range(1, 5):
            self._add('%s/highway_%d/kernel' % (scope, i), (128, 256))   # [H | T]
            self._add('%s/highway_%d/bias' % (scope, i), (256,))
        self._add(scope + '/bigru/wx', (128, 768))    # fw (r|u|c) | bw (r|u|c)
        self._add(scope + '/bigru/bias', (768,))
        for d in ('fw', 'bw'):
            self._add('%s/bigru/%s_whg' % (scope, d), (128, 256))
            self._add('%s/bigru/%s_whc' % (scope, d), (128, 128))

    # ---- self-description (checkpoints) --------------------------------------------------------------------
    TF_SCOPE = 'model/inference'      # reference train.py:101 opens 'model', tacotron.py:35 'inference'; synthesizer.py:25 reads
                                      # 'model/inference/embedding_id' from a checkpoint to recover the speaker count

    def describe(self):
        """[[name, offset, shape], ...] of the trainable entries followed by the moving statistics ('bn:' prefix): the named
        layout a checkpoint of the flat buffers stores, so that a file is self-describing and a changed layout is refused."""
        return ([[e.name, int(e.offset), [int(d) for d in e.shape]] for e in self.entries.values()] +
                [['bn:' + e.name, int(e.offset), [int(d) for d in e.shape]] for e in self.bn_entries.values()])

    def signature(self):
        import hashlib
        return hashlib.sha256(repr(self.describe()).encode()).hexdigest()

    # ---- views -----------------------------------------------------------------------------------
    def view(self, flat, name):
        e = self.entries[name]
        return flat[e.offset:e.offset + e.size].view(*e.shape)

    def bnview(self, flat, name):
        e = self.bn_entries[name]
        return flat[e.offset:e.offset + e.size].view(*e.shape)

    # ---- TF-style named tensors <-> flat layout -------------------------------------------------------
    def _cbhg_K(self, scope):
        return 16 if scope == 'encoder_cbhg' else 8

    def load_named(self, named, flat, bnflat):
        """named: {tf-style name: array} as produced by oracle init / a converted checkpoint.
        flat / bnflat: torch tensors (any device) to fill."""
        import torch

        def put(name, arr):
            v = self.view(flat, name)
            v.copy_(torch.as_tensor(np.ascontiguousarray(arr), dtype=flat.dtype).reshape(v.shape))

        def putbn(name, arr):
            v = self.bnview(bnflat, name)
            v.copy_(torch.as_tensor(np.ascontiguousarray(arr), dtype=bnflat.dtype).reshape(v.shape))

        g = lambda k: np.asarray(named[k])
        put('embedding', g('embedding'))
        if self.Es:
            put('embedding_id', g('embedding_id'))
        for sc in ('prenet/dense_1', 'prenet/dense_2', 'decoder_prenet/dense_1', 'decoder_prenet/dense_2',
                   'concat_projection', 'output_projection'):
            put(sc + '/kernel', g(sc + '/kernel'))
            put(sc + '/bias', g(sc + '/bias'))
        for sc in ('attention/memory_layer/kernel', 'attention/query_layer/kernel', 'attention/attention_v'):
            put(sc, g(sc))
        for sc, n_in, n in (('attention_gru', 128, 256), ('decoder_gru_1', 256, 256), ('decoder_gru_2', 256, 256)):
            gk, ck = g(sc + '/gates/kernel'), g(sc + '/candidate/kernel')
            put(sc + '/wx', np.concatenate([gk[:n_in], ck[:n_in]], axis=1))
            put(sc + '/bias', np.concatenate([g(sc + '/gates/bias'), g(sc + '/candidate/bias')]))
            put(sc + '/whg', gk[n_in:])
            put(sc + '/whc', ck[n_in:])
        for sc in ('encoder_cbhg', 'post_cbhg'):
            K = self._cbhg_K(sc)
            bank = ['%s/conv_bank/conv1d_%d' % (sc, k) for k in range(1, K + 1)]
            put(sc + '/conv_bank/kernel', np.concatenate([g(b + '/kernel') for b in bank], axis=0))
            for s in ('bias', 'gamma', 'beta'):
                put(sc + '/conv_bank/' + s, np.concatenate([g(b + '/' + s) for b in bank]))
            for s in ('moving_mean', 'moving_variance'):
                putbn(sc + '/conv_bank/' + s, np.concatenate([g(b + '/' + s) for b in bank]))
            for pj in ('proj_1', 'proj_2'):
                for s in ('kernel', 'bias', 'gamma', 'beta'):
                    put('%s/%s/%s' % (sc, pj, s), g('%s/%s/%s' % (sc, pj, s)))
                for s in ('moving_mean', 'moving_variance'):
                    putbn('%s/%s/%s' % (sc, pj, s), g('%s/%s/%s' % (sc, pj, s)))
            if sc + '/highway_dense/kernel' in self.entries:
                put(sc + '/highway_dense/kernel', g(sc + '/highway_dense/kernel'))
                put(sc + '/highway_dense/bias', g(sc + '/highway_dense/bias'))
            for i in range(1, 5):
                hs = '%s/highway_%d' % (sc, i)
                put(hs + '/kernel', np.concatenate([g(hs + '/H/kernel'), g(hs + '/T/kernel')], axis=1))
                put(hs + '/bias', np.concatenate([g(hs + '/H/bias'), g(hs + '/T/bias')]))
            wx, bx = [], []
            for d in ('fw', 'bw'):
                gk, ck = g('%s/gru_%s/gates/kernel' % (sc, d)), g('%s/gru_%s/candidate/kernel' % (sc, d))
                wx += [gk[:128], ck[:128]]
                bx += [g('%s/gru_%s/gates/bias' % (sc, d)), g('%s/gru_%s/candidate/bias' % (sc, d))]
                put('%s/bigru/%s_whg' % (sc, d), gk[128:])
                put('%s/bigru/%s_whc' % (sc, d), ck[128:])
            put(sc + '/bigru/wx', np.concatenate(wx, axis=1))
            put(sc + '/bigru/bias', np.concatenate(bx))
        lk = np.zeros((256, self.ld_lin), dtype=np.float64)
        lk[:, :self.num_freq] = g('linear/kernel')
        lb = np.zeros(self.ld_lin, dtype=np.float64)
        lb[:self.num_freq] = g('linear/bias')
        put('linear/kernel', lk)
        put('linear/bias', lb)

    def export_named(self, flat, bnflat=None):
        """flat layout -> {tf-style name: np.ndarray} (inverse of load_named).  Works for params, grads, slots."""
        out = OrderedDict()
        V = lambda n: self.view(flat, n).detach().cpu().numpy()
        out['embedding'] = V('embedding')
        if self.Es:
            out['embedding_id'] = V('embedding_id')
        for sc in ('prenet/dense_1', 'prenet/dense_2'):
            out[sc + '/kernel'], out[sc + '/bias'] = V(sc + '/kernel'), V(sc + '/bias')

        def cbhg(sc):
            K = self._cbhg_K(sc)
            bk = V(sc + '/conv_bank/kernel')
            off = 0
            for k in range(1, K + 1):
                b = '%s/conv_bank/conv1d_%d' % (sc, k)
                out[b + '/kernel'] = bk[off:off + k]
                off += k
                for s in ('bias', 'gamma', 'beta'):
                    out[b + '/' + s] = V(sc + '/conv_bank/' + s)[(k - 1) * 128:k * 128]
                if bnflat is not None:
                    for s in ('moving_mean', 'moving_variance'):
                        out[b + '/' + s] = self.bnview(bnflat, sc + '/conv_bank/' + s).cpu().numpy()[(k - 1) * 128:k * 128]
            for pj in ('proj_1', 'proj_2'):
                for s in ('kernel', 'bias', 'gamma', 'beta'):
                    out['%s/%s/%s' % (sc, pj, s)] = V('%s/%s/%s' % (sc, pj, s))
                if bnflat is not None:
                    for s in ('moving_mean', 'moving_variance'):
                        out['%s/%s/%s' % (sc, pj, s)] = self.bnview(bnflat, '%s/%s/%s' % (sc, pj, s)).cpu().numpy()
            if sc + '/highway_dense/kernel' in self.entries:
                out[sc + '/highway_dense/kernel'] = V(sc + '/highway_dense/kernel')
                out[sc + '/highway_dense/bias'] = V(sc + '/highway_dense/bias')
            for i in range(1, 5):
                hs = '%s/highway_%d' % (sc, i)
                w, b = V(hs + '/kernel'), V(hs + '/bias')
                out[hs + '/H/kernel'], out[hs + '/H/bias'] = w[:, :128], b[:128]
                out[hs + '/T/kernel'], out[hs + '/T/bias'] = w[:, 128:], b[128:]
            wx, bx = V(sc + '/bigru/wx'), V(sc + '/bigru/bias')
            for di, d in enumerate(('fw', 'bw')):
                o = di * 384
                whg, whc = V('%s/bigru/%s_whg' % (sc, d)), V('%s/bigru/%s_whc' % (sc, d))
                out['%s/gru_%s/gates/kernel' % (sc, d)] = np.concatenate([wx[:, o:o + 256], whg], axis=0)
                out['%s/gru_%s/gates/bias' % (sc, d)] = bx[o:o + 256]
                out['%s/gru_%s/candidate/kernel' % (sc, d)] = np.concatenate([wx[:, o + 256:o + 384], whc], axis=0)
                out['%s/gru_%s/candidate/bias' % (sc, d)] = bx[o + 256:o + 384]

        cbhg('encoder_cbhg')
        for sc in ('attention/memory_layer/kernel', 'attention/query_layer/kernel', 'attention/attention_v'):
            out[sc] = V(sc)
        for sc in ('decoder_prenet/dense_1', 'decoder_prenet/dense_2'):
            out[sc + '/kernel'], out[sc + '/bias'] = V(sc + '/kernel'), V(sc + '/bias')

        def gru(sc, n):
            wx, b = V(sc + '/wx'), V(sc + '/bias')
            out[sc + '/gates/kernel'] = np.concatenate([wx[:, :2 * n], V(sc + '/whg')], axis=0)
            out[sc + '/gates/bias'] = b[:2 * n]
            out[sc + '/candidate/kernel'] = np.concatenate([wx[:, 2 * n:], V(sc + '/whc')], axis=0)
            out[sc + '/candidate/bias'] = b[2 * n:]

        gru('attention_gru', 256)
        out['concat_projection/kernel'], out['concat_projection/bias'] = V('concat_projection/kernel'), V('concat_projection/bias')
        gru('decoder_gru_1', 256)
        gru('decoder_gru_2', 256)
        out['output_projection/kernel'], out['output_projection/bias'] = V('output_projection/kernel'), V('output_projection/bias')
        cbhg('post_cbhg')
        out['linear/kernel'] = V('linear/kernel')[:, :self.num_freq]
        out['linear/bias'] = V('linear/bias')[:self.num_freq]
        return out


# ---- TF initialisers for a fresh model -------------------------------------------------
def _glorot(rng, shape):
    if len(shape) == 1:
        fi = fo = shape[0]
    elif len(shape) == 2:
        fi, fo = shape
    else:
        rf = int(np.prod(shape[:-2]))
        fi, fo = shape[-2] * rf, shape[-1] * rf
    lim = np.sqrt(6.0 / (fi + fo))
    return rng.uniform(-lim, lim, size=shape)


def _trunc_normal(rng, shape, std):
    x = rng.normal(size=shape)
    bad = np.abs(x) > 2
    while bad.any():
        x[bad] = rng.normal(size=int(bad.sum()))
        bad = np.abs(x) > 2
    return x * std


def init_named(layout, seed=0):
    """Fresh TF-style variables (tf.truncated_normal(0.5) embeddings, glorot-uniform kernels, zero biases,
    GRU gate bias 1.0, highway T bias -1.0; reference tacotron.py:44,51, modules.py:89) in creation order."""
    rng = np.random.RandomState(seed)
    P = OrderedDict()
    r, nm, nf = layout.r, layout.num_mels, layout.num_freq

    def convbn(sc, k, cin, cout):
        P[sc + '/kernel'] = _glorot(rng, (k, cin, cout))
        P[sc + '/bias'] = np.zeros(cout)
        P[sc + '/gamma'] = np.ones(cout)
        P[sc + '/beta'] = np.zeros(cout)
        P[sc + '/moving_mean'] = np.zeros(cout)
        P[sc + '/moving_variance'] = np.ones(cout)

    def gru(sc, n_in, n):
        P[sc + '/gates/kernel'] = _glorot(rng, (n_in + n, 2 * n))
        P[sc + '/gates/bias'] = np.ones(2 * n)
        P[sc + '/candidate/kernel'] = _glorot(rng, (n_in + n, n))
        P[sc + '/candidate/bias'] = np.zeros(n)

    def dense(sc, cin, cout):
        P[sc + '/kernel'] = _glorot(rng, (cin, cout))
        P[sc + '/bias'] = np.zeros(cout)

    def cbhg(sc, K, cin, proj):
        for k in range(1, K + 1):
            convbn('%s/conv_bank/conv1d_%d' % (sc, k), k, cin, 128)
        convbn(sc + '/proj_1', 3, K * 128, proj[0])
        convbn(sc + '/proj_2', 3, proj[0], proj[1])
        if proj[1] != 128:
            dense(sc + '/highway_dense', proj[1], 128)
        for i in range(1, 5):
            dense('%s/highway_%d/H' % (sc, i), 128, 128)
            P['%s/highway_%d/T/kernel' % (sc, i)] = _glorot(rng, (128, 128))
            P['%s/highway_%d/T/bias' % (sc, i)] = np.full(128, -1.0)
        gru(sc + '/gru_fw', 128, 128)
        gru(sc + '/gru_bw', 128, 128)

    P['embedding'] = _trunc_normal(rng, (layout.vocab, layout.Et), 0.5)
    if layout.Es:
        P['embedding_id'] = _trunc_normal(rng, (layout.id_num, layout.Es), 0.5)
    dense('prenet/dense_1', layout.Et + layout.Es, 256)
    dense('prenet/dense_2', 256, 128)
    cbhg('encoder_cbhg', 16, 128, (128, 128))
    P['attention/memory_layer/kernel'] = _glorot(rng, (256, 256))
    P['attention/query_layer/kernel'] = _glorot(rng, (256, 256))
    P['attention/attention_v'] = _glorot(rng, (256,))
    dense('decoder_prenet/dense_1', nm + 256, 256)
    dense('decoder_prenet/dense_2', 256, 128)
    gru('attention_gru', 128, 256)
    dense('concat_projection', 512, 256)
    gru('decoder_gru_1', 256, 256)
    gru('decoder_gru_2', 256, 256)
    dense('output_projection', 256, nm * r)
    cbhg('post_cbhg', 8, nm, (256, nm))
    dense('linear', 256, nf)
    return P
