"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C-ABI library, against the
float64 oracle and the committed golden fixtures.  Tolerance: 1e-3 relative on mel / linear outputs
(BASELINE.json north_star); gradients and the post-step parameters are held to the same relative bar
(observed ~1e-5).  'Relative' = max|a-b| / max|b| per tensor."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden')
TOL = 1e-3


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def dev_batch(b, dev):
    t = lambda k, dt: torch.tensor(b[k], device=dev, dtype=dt) if b.get(k) is not None else None
    return (t('inputs', torch.int32), t('input_lengths', torch.int32), t('mel_targets', torch.float32),
            t('linear_targets', torch.float32), t('identities', torch.int32))


def run_engine_step(P, b, r, idn, apply=True, regularity=None):
    from tacotron_multispeaker_amd.engine import Engine
    eng = Engine(id_num=idn, r=r, named_params=P)
    if regularity:
        eng.set_regularity(**regularity)
    i, l, m, lin, ids = dev_batch(b, eng.dev)
    eng.forward(i, l, m, ids, linear_targets=lin)          # the call sequence of Engine.train_step
    eng.loss(lin)
    eng.backward()
    torch.cuda.synchronize()
    eng.check_errors()                 # a timed-out cluster hand-off would make everything below meaningless
    out = dict(mel=eng.mel_outputs.cpu().numpy(), lin=eng.linear_outputs.cpu().numpy(), align=eng.alignments.cpu().numpy(),
               loss=eng.loss_values(), grads=eng.export_named('grads'), eng=eng)
    if apply:
        eng.optimizer_step()
        torch.cuda.synchronize()
        out['params'] = eng.export_named('params')
        out['info'] = eng.info.cpu().numpy()
        out['step'] = int(eng.global_step.item())
    return out


@pytest.mark.parametrize('name', ['tiny', 'multi_r2'])
def test_training_step_matches_golden(name):
    from oracle import tacotron_np as onp
    g = np.load(os.path.join(GOLD, 'step_%s.npz' % name))
    N, Ti, To, r, idn, ps, bs = [int(x) for x in g['config']]
    P = onp.init_params(seed=ps, r=r, id_num=idn)
    b = onp.synth_batch(N, Ti, To, r, seed=bs, id_num=idn)
    o = run_engine_step(P, b, r, idn)
    assert rel(o['mel'], g['mel_outputs']) < TOL
    assert rel(o['lin'], g['linear_outputs']) < TOL
    assert rel(o['align'], g['alignments']) < TOL
    assert abs(o['loss'][0] - g['loss'][0]) < 1e-5 * g['loss'][0]
    assert abs(o['info'][0] - float(g['global_norm'])) < 1e-4 * float(g['global_norm'])
    assert abs(o['info'][1] - float(g['learning_rate'])) < 1e-6 * float(g['learning_rate'])
    gmax = max(g['grad_l2'])
    for n, l2, s in zip(g['grad_names'], g['grad_l2'], g['grad_sum']):
        mine = o['grads'][str(n)].astype(np.float64)
        assert abs(np.sqrt((mine ** 2).sum()) - l2) < TOL * l2 + 1e-7 * gmax, n      # bias-before-BN grads are exactly 0
    for n, s, a in zip(g['param_names'], g['param_sum_after'], g['param_abs_after']):
        mine = o['params'][str(n)].astype(np.float64)
        assert abs(mine.sum() - s) < 1e-5 * a + 1e-6, n
    assert o['step'] == 1


@pytest.mark.parametrize('cfg', [(4, 48, 120, 5, 0, 0), (5, 17, 35, 5, 0, 0), (1, 9, 12, 3, 0, 0), (3, 33, 16, 1, 7, 0),
                                 (2, 40, 64, 2, 3, 0), (4, 24, 40, 5, 0, 1), (4, 48, 120, 5, 0, 2)])
def test_training_step_matches_oracle(cfg):
    """ragged / odd batch sizes, r in {1,2,3,5}, single- and multi-speaker, perturbed BN/bias parameters.
    Padded text positions all carry embedding[0], so the conv-bank outputs are EXACTLY equal over time there in the HIP
    path (max-pool ties at non-zero values, routed to the first max like TF CPU MaxPoolGrad; the oracle's pooling is
    tie-aware for the same reason).  Mode 0 fills the padded text positions with random ids, mode 2 keeps the feeder's
    real zero padding, mode 1 adds an EOS-only row; all three are held to 1e-3 per gradient tensor on the linear piece
    the HIP path took (tests/decisions.py)."""
    from oracle import tacotron_np as onp
    N, Ti, To, r, idn, mode = cfg
    eos_only = mode == 1
    P = onp.init_params(seed=21, r=r, id_num=idn)
    rng = np.random.RandomState(5)
    for k in P:                                   # move biases / BN affine away from their trivial initial values
        if k.endswith(('/bias', '/beta')):
            P[k] = P[k] + 0.1 * rng.standard_normal(P[k].shape)
        if k.endswith('/gamma'):
            P[k] = P[k] * (1 + 0.2 * rng.standard_normal(P[k].shape))
    b = onp.synth_batch(N, Ti, To, r, seed=31, id_num=idn)
    if mode == 0:
        pad = b['inputs'] == 0
        b['inputs'][pad] = np.random.RandomState(7).randint(2, 7352, size=int(pad.sum()))
    if eos_only:
        b['input_lengths'][0] = 1                 # shortest possible text: EOS only
        b['inputs'][0, :] = 0; b['inputs'][0, 0] = 1
    _oracle_step_compare(P, b, r, idn)


@pytest.mark.parametrize('env,cfg', [({'TACO_XCD_LOCAL': '0'}, (3, 33, 16, 1, 7)),       # odd batch, 4 pipeline chunks of 4 steps
                                     ({'TACO_XCD_LOCAL': '0'}, (5, 40, 120, 5, 0)),      # odd batch, S = 24
                                     ({'TACO_OVERLAP_WGRAD': '0'}, (4, 48, 120, 5, 0)),   # weight gradients on the main stream
                                     ({'TACO_FLUSH_AT': '0'}, (4, 48, 120, 5, 0))])       # post-net dW released after the FIRST chunk
def test_training_step_on_the_alternative_paths_matches_oracle(env, cfg, monkeypatch):
    """Paths a default run does not take, each a full training step against the float64 oracle:
    * TACO_XCD_LOCAL=0: every persistent cluster keeps the agent-scope granule form (`global_store sc1`) -- the
      placement-independent hand-off a cluster falls back to when its members are not on one XCD (csrc/xcd_granule.hpp); the
      placement counters stay 0 because no cluster runs the check, while a default engine counts every cluster launch;
    * TACO_OVERLAP_WGRAD=0 / TACO_FLUSH_AT=0: the output-projection weight gradient reads the whole dOUT that the post-net bank's
      input-gradient pieces accumulate into on another stream -- it must wait for the last piece wherever it is emitted."""
    from oracle import tacotron_np as onp
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    N, Ti, To, r, idn = cfg
    P = onp.init_params(seed=27, r=r, id_num=idn)
    rng = np.random.RandomState(8)
    for k in P:
        if k.endswith(('/bias', '/beta')):
            P[k] = P[k] + 0.1 * rng.standard_normal(P[k].shape)
    b = onp.synth_batch(N, Ti, To, r, seed=35, id_num=idn)
    o, _ = _oracle_step_compare(P, b, r, idn)
    words = [int(x) for x in o['eng'].err.cpu().tolist()]
    assert words[0] == 0
    if 'TACO_XCD_LOCAL' in env:
        assert words[1] == 0 and words[2] == 0
    else:
        assert words[2] > 0 and 0 <= words[1] <= words[2]


@pytest.mark.parametrize('cfg,plan', [((4, 48, 120, 5, 0), '0.625:0.775:0.9'),     # chunks (0,72) (72,96) (96,104) (104,120)
                                      ((3, 20, 135, 3, 2), '0.55:0.8'),            # odd T_out, bands that are not multiples of 32
                                      ((2, 12, 70, 2, 0), '0.5:0.7:0.95')])        # first cut exactly at the middle: no band yet
def test_post_net_bigru_frame_band_pipelines_match_oracle(cfg, plan, monkeypatch):
    """The post-net biGRU cut into chunk launches (state carried through a buffer) with the consumers of the frames each chunk
    completes running beside the next chunk: linear layer + L1 loss + linear input gradient after the forward pass
    (taco_dense_rows_fwd, taco_l1_loss_rows, taco_dense_rows_bwd_data), input-projection gradient + highway-stack BPTT + 80->128
    dense gradient after the backward pass (taco_highway4_bwd_rows).  BASELINE shapes enable it from T_out = 256; here it is forced
    on small shapes and held, like every other path, to the float64 oracle: outputs, loss, every gradient tensor, the update."""
    from oracle import tacotron_np as onp
    monkeypatch.setenv('TACO_TAIL_MIN_T', '16')
    monkeypatch.setenv('TACO_TAIL_PLAN', plan)
    N, Ti, To, r, idn = cfg
    from tacotron_multispeaker_amd.engine import Engine
    monkeypatch.setattr(Engine, 'FUSED_HIGHWAY_MIN_ROWS', 1)    # the banded BPTT uses the fused highway kernel
    assert len(Engine._tail_chunks(None, To, True)) >= 3
    P = onp.init_params(seed=41, r=r, id_num=idn)
    rng = np.random.RandomState(4)
    for k in P:
        if k.endswith(('/bias', '/beta')):
            P[k] = P[k] + 0.1 * rng.standard_normal(P[k].shape)
    b = onp.synth_batch(N, Ti, To, r, seed=51, id_num=idn)
    _oracle_step_compare(P, b, r, idn)


@pytest.mark.parametrize('cfg', [(66, 14, 20, 5, 0), (2, 300, 30, 5, 0), (130, 10, 15, 5, 3)])
def test_shapes_beyond_one_cluster_launch_match_oracle(cfg):
    """Limits of the persistent cluster kernels: one attention launch holds <= 64 batch rows (8 workgroups per 2 rows on
    256 CUs) and T_in up to ~270 (key / memory tiles in LDS), one GRU(256) launch <= 128 rows.  Larger batches run in row
    blocks (N = 66: 64 + 2, N = 130: 64 + 64 + 2 / 128 + 2); longer inputs run on the per-step kernels (T_in = 300).
    Same bars as everywhere else: outputs 1e-3 (observed ~1e-6), every gradient tensor 1e-3 on the linear piece the HIP
    path took (tests/decisions.py)."""
    from oracle import tacotron_np as onp
    from tacotron_multispeaker_amd._lib import lib
    N, Ti, To, r, idn = cfg
    assert not lib.load().taco_attn_cluster_supported(N, Ti)
    P = onp.init_params(seed=29, r=r, id_num=idn)
    b = onp.synth_batch(N, Ti, To, r, seed=41, id_num=idn)
    _oracle_step_compare(P, b, r, idn, apply=False, gabs=2e-5)


def _oracle_step_compare(P, b, r, idn, gtol=TOL, apply=True, gabs=1e-6, regularity=None):
    """Full step on the HIP path vs the float64 oracle: outputs / alignments / loss at 1e-3 relative (observed ~1e-6), every
    gradient tensor at `gtol` relative L2 (+ gabs of the largest gradient norm: the conv biases in front of a batch norm have
    exactly zero gradient), global norm and post-step parameters.  Gradients are compared on the linear piece the HIP path
    took: see tests/decisions.py (every decision that differs from the oracle's must be an fp32-rounding near-tie)."""
    from decisions import oracle_step_aligned

    def run():
        o = run_engine_step(P, b, r, idn, apply=apply, regularity=regularity)
        return o['eng'], o
    ts, last, o, flips = oracle_step_aligned(P, b, r, idn, run, regularity=regularity)
    info = ts.apply(last)
    assert rel(o['mel'], last['out']['mel_outputs'].detach().numpy()) < TOL
    assert rel(o['lin'], last['out']['linear_outputs'].detach().numpy()) < TOL
    assert rel(o['align'], last['out']['alignments'].detach().numpy()) < TOL
    assert abs(o['loss'][0] - last['loss']) < 1e-5 * last['loss']
    gmax = max(float(v.norm()) for v in last['grads'].values())
    e2 = n2 = 0.0
    for k, v in last['grads'].items():
        v = v.numpy()
        err = np.sqrt(((o['grads'][k] - v) ** 2).sum())
        assert err < gtol * np.sqrt((v ** 2).sum()) + gabs * gmax, k
        e2 += err ** 2; n2 += (v ** 2).sum()
    assert np.sqrt(e2 / n2) < 1e-4                  # all gradients together (observed ~3e-6)
    if apply:
        assert abs(o['info'][0] - info['global_norm']) < 1e-4 * info['global_norm']
        for k, v in ts.P.items():
            assert np.abs(o['params'][k] - v.detach().numpy()).max() < 1e-5, k
        assert o['step'] == 1
    return o, last


# (N, T_in, T_out, r, id_num, force register-weights variant, expected BPTT variant)
@pytest.mark.parametrize('cfg', [(4, 200, 96, 2, 5, False, 2),      # C5-shaped: T_in 200, r 2, S 48 (4 pipeline chunks), multispeaker
                                 (2, 192, 120, 5, 0, False, 2),     # C2x-shaped: T_in 192, r 5, S 24
                                 (2, 152, 40, 5, 0, False, 2),      # just past the LDS-weights limit (T_in 148)
                                 (3, 256, 45, 5, 3, False, 2),      # near the LDS limit of the cluster path (~270)
                                 (2, 148, 40, 5, 0, False, 1),      # last T_in of the LDS-weights variant (prefetch wave + stage in LDS)
                                 (4, 40, 100, 5, 0, True, 2)])      # register-weights variant forced at a small T_in
def test_attention_bptt_variants_long_inputs(cfg, monkeypatch):
    """attn_cluster_bwd_launch runs attn_cluster_bwd_k<true> (prenet-gradient weight slices in LDS) while they fit beside the
    key / memory tiles (T_in <= 148) and attn_cluster_bwd_k<false> (all weight slices in registers) for T_in 149..~270 -- the
    kernel BASELINE configs C5 (T_in 200, r 2, max_iters 400) and C2x (T_in 192) use.  Full training step against the
    float64 oracle; the chunk pipeline is active (S >= 8)."""
    from oracle import tacotron_np as onp
    from tacotron_multispeaker_amd._lib import lib
    N, Ti, To, r, idn, force, variant = cfg
    if force:
        monkeypatch.setenv('TACO_ATTN_NO_WLDS', '1')
    assert lib.load().taco_attn_cluster_supported(N, Ti) == 1
    assert lib.load().taco_attn_cluster_bwd_variant(N, Ti) == variant
    P = onp.init_params(seed=33, r=r, id_num=idn)
    rng = np.random.RandomState(6)
    for k in P:
        if k.endswith(('/bias', '/beta')):
            P[k] = P[k] + 0.1 * rng.standard_normal(P[k].shape)
    b = onp.synth_batch(N, Ti, To, r, seed=43, id_num=idn)
    _oracle_step_compare(P, b, r, idn)


@pytest.mark.parametrize('name,N,Ti,To,r,idn', [('C4', 32, 64, 480, 5, 460), ('C5', 16, 200, 800, 2, 460)])
def test_full_size_multispeaker_configs(name, N, Ti, To, r, idn):
    """BASELINE configs 4 and 5 at their real per-GPU shape: forward + loss against the fp32 CPU
    restatement, then the size-independent properties of a full training step: alignments are distributions over ALL T_in,
    padded encoder rows are exactly zero, no cluster hand-off timed out, the loss and every gradient are finite, the global
    norm the optimizer saw equals the norm of the exported gradients, and a second step lowers the loss."""
    from oracle import tacotron_np as onp, tacotron_torch as ot
    from tacotron_multispeaker_amd.engine import Engine
    P = onp.init_params(seed=0, r=r, id_num=idn)
    b = onp.synth_batch(N, Ti, To, r, seed=1234, id_num=idn)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    Pt = ot.to_torch(P, torch.float32, requires_grad=False)
    with torch.no_grad():
        ref = ot.forward(Pt, b['inputs'], b['input_lengths'], torch.tensor(b['mel_targets']), b['identities'], idn, r)
        ref_loss = float(ot.loss_fn(ref['mel_outputs'], ref['linear_outputs'], torch.tensor(b['mel_targets']),
                                    torch.tensor(b['linear_targets']))[0])
    eng = Engine(r=r, id_num=idn, named_params=P, init_lr=0.002, decay_lr=False, tf_sparse_norm=False)
    i, l, m, lin, ids = dev_batch(b, eng.dev)
    eng.train_step(i, l, m, lin, ids)
    torch.cuda.synchronize()
    eng.check_errors()
    assert rel(eng.mel_outputs.cpu().numpy(), ref['mel_outputs'].numpy()) < TOL
    assert rel(eng.linear_outputs.cpu().numpy(), ref['linear_outputs'].numpy()) < TOL
    assert rel(eng.alignments.cpu().numpy(), ref['alignments'].numpy()) < TOL
    first = eng.loss_values()[0]
    assert abs(first - ref_loss) < 1e-4 * ref_loss
    al = eng.alignments.cpu().numpy()
    assert al.shape == (N, Ti, To // r) and np.abs(al.sum(axis=1) - 1).max() < 1e-5 and al.min() >= 0
    enc = eng.encoder_outputs.cpu().numpy()
    for n in range(N):
        assert np.all(enc[n, b['input_lengths'][n]:] == 0)
    g = eng.grads.cpu().numpy().astype(np.float64)
    assert np.all(np.isfinite(g))
    assert abs(float(eng.info[0].item()) - np.sqrt((g ** 2).sum())) < 1e-4 * np.sqrt((g ** 2).sum())
    eng.train_step(i, l, m, lin, ids)
    torch.cuda.synchronize()
    eng.check_errors()
    second = eng.loss_values()[0]
    assert np.isfinite(second) and second < first and int(eng.global_step.item()) == 2
    del eng
    # and the whole step -- every gradient tensor, the global norm, the post-step parameters -- against the FLOAT64 oracle at the
    # full shape: C5 is the longest BPTT chain of BASELINE.json (S = 400 decoder steps, register-weight attention variants, four
    # pipeline chunks of 100 steps), where fp32 error accumulates most
    _oracle_step_compare(P, b, r, idn)


def test_full_size_c2_training_step_matches_float64_oracle():
    """BASELINE config 2 (N=32, T_in=128, T_out=640, r=5), the benchmarked workload: the whole training step (outputs, loss,
    every gradient tensor, global norm, post-step parameters) against the float64 oracle, real zero-padded text included."""
    from oracle import tacotron_np as onp
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    P = onp.init_params(seed=0, r=5)
    b = onp.synth_batch(32, 128, 640, 5, seed=1234)
    _oracle_step_compare(P, b, 5, 0)


def test_full_size_c2_with_the_forward_gemms_on_the_bf16_pipe(monkeypatch):
    """TACO_X3=2 (NOT the default): the forward GEMMs of the large layers multiply with three bf16 MFMAs per fp32 product as well.  Same
    comparison as above at the same tolerances for outputs, loss, every gradient tensor, norm and update; only the decision test's
    notion of a near-tie follows the arithmetic: a forward product good to 2^-17 moves a pre-activation by ~3e-6 of the site's RMS (fp32:
    5e-7), so flipped ReLU / max-pool decisions are allowed up to 1e-4 of it (default mode: 3e-5; measured here: up to 3.7e-5) and at
    6x the fp32 rate."""
    import decisions
    from oracle import tacotron_np as onp
    monkeypatch.setenv('TACO_X3', '2')
    monkeypatch.setattr(decisions, 'NEAR', 1e-4)
    monkeypatch.setattr(decisions, 'RATE', 1.2e-4)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    P = onp.init_params(seed=0, r=5)
    b = onp.synth_batch(32, 128, 640, 5, seed=1234)
    _oracle_step_compare(P, b, 5, 0)


@pytest.mark.parametrize('cfg', [(3, 20, 225, 5, 0, False), (2, 12, 135, 3, 2, True), (4, 16, 205, 5, 0, True)])
def test_alignment_regularisers_match_oracle(cfg, monkeypatch):
    """loss_regularity of tacotron.py:140-171 (second softmax over the decoder steps, one-order
    dynamics, first-word mass after step 40, row-sum variance, entropy) and its gradient through the attention BPTT, on
    the cluster kernels and on the per-step fallback kernels.  Weights are large enough that the regulariser gradient is
    a visible share of every attention-side gradient.  The one-order term is a sum of |pr[s] - pr[s+1]| whose gradient is a
    sign: where two neighbouring probabilities agree to fp32 rounding the sign differs from the float64 oracle's, so the
    gradients are held to 5e-3 here (values and the kink-free terms are exact to 1e-4 / 1e-5)."""
    from oracle import tacotron_np as onp, tacotron_torch as ot
    from tacotron_multispeaker_amd.engine import Engine
    N, Ti, To, r, idn, per_step = cfg
    if per_step:
        monkeypatch.setenv('TACO_NO_CLUSTER', '1')
    reg = dict(overwrought=0.02, oneorder_dynamic=0.01, variance_between_row=0.003, alignment_entropy=5.0)
    P = onp.init_params(seed=23, r=r, id_num=idn)
    b = onp.synth_batch(N, Ti, To, r, seed=37, id_num=idn)
    pad = b['inputs'] == 0
    b['inputs'][pad] = np.random.RandomState(9).randint(2, 7352, size=int(pad.sum()))     # tie-free max-pool (see above)
    from decisions import oracle_step_aligned
    plain = ot.TrainState(P, torch.float64, id_num=idn, r=r).forward_backward(b)

    def run():
        o = run_engine_step(P, b, r, idn, apply=False, regularity=reg)
        return o['eng'], o
    ts, last, o, flips = oracle_step_aligned(P, b, r, idn, run, regularity=reg)       # ReLU / max-pool decisions aligned
    eng = o['eng']
    loss = o['loss'][0]
    a_np = last['out']['alignments'].detach().numpy()
    assert abs(onp.alignment_regularity(a_np, **reg) - last['loss_regularity']) < 1e-9 * abs(last['loss_regularity'])
    assert abs(eng.loss_regularity - last['loss_regularity']) < 1e-4 * abs(last['loss_regularity'])
    assert abs(loss - last['loss']) < 1e-5 * last['loss']
    grads = o['grads']
    gmax = max(float(v.norm()) for v in last['grads'].values())
    moved = 0
    for k, v in last['grads'].items():
        v = v.numpy()
        err = np.sqrt(((grads[k] - v) ** 2).sum())
        assert err < 5e-3 * np.sqrt((v ** 2).sum()) + 1e-6 * gmax, k
        if np.sqrt(((plain['grads'][k].numpy() - v) ** 2).sum()) > 0.05 * np.sqrt((v ** 2).sum()):
            moved += 1
    assert moved >= 5          # the regulariser really changes the attention / encoder gradients


def test_full_size_c2_forward_matches_cpu_restatement():
    """BASELINE config 2 (N=32, T_in=128, T_out=640, r=5) forward + loss against the fp32 CPU restatement."""
    from oracle import tacotron_np as onp, tacotron_torch as ot
    from tacotron_multispeaker_amd.engine import Engine
    N, Ti, To, r = 32, 128, 640, 5
    P = onp.init_params(seed=0, r=r)
    b = onp.synth_batch(N, Ti, To, r, seed=1234)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    Pt = ot.to_torch(P, torch.float32, requires_grad=False)
    with torch.no_grad():
        ref = ot.forward(Pt, b['inputs'], b['input_lengths'], torch.tensor(b['mel_targets']), None, 0, r)
        ref_loss = float(ot.loss_fn(ref['mel_outputs'], ref['linear_outputs'], torch.tensor(b['mel_targets']),
                                    torch.tensor(b['linear_targets']))[0])
    eng = Engine(r=r, named_params=P)
    i, l, m, lin, _ = dev_batch(b, eng.dev)
    eng.forward(i, l, m)
    eng.loss(lin)
    torch.cuda.synchronize()
    assert rel(eng.mel_outputs.cpu().numpy(), ref['mel_outputs'].numpy()) < TOL
    assert rel(eng.linear_outputs.cpu().numpy(), ref['linear_outputs'].numpy()) < TOL
    assert rel(eng.alignments.cpu().numpy(), ref['alignments'].numpy()) < TOL
    assert abs(eng.loss_values()[0] - ref_loss) < 1e-4 * ref_loss
    # size-independent properties: alignments are distributions over ALL T_in; padded encoder rows are zero
    al = eng.alignments.cpu().numpy()
    assert np.abs(al.sum(axis=1) - 1).max() < 1e-5 and al.min() >= 0
    enc = eng.encoder_outputs.cpu().numpy()
    for n in range(N):
        assert np.all(enc[n, b['input_lengths'][n]:] == 0)


def test_full_size_training_reduces_loss_and_replays_from_graph():
    from tacotron_multispeaker_amd.engine import Engine
    from tacotron_multispeaker_amd import synth
    eng = Engine(r=5, seed=0, init_lr=0.002, decay_lr=False)
    args = synth.batch_to_device(synth.synth_batch(32, 128, 640, 5, seed=1234), eng.dev)
    eng.train_step(*args[:4])
    first = eng.loss_values()[0]
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        eng.train_step(*args[:4])
    for _ in range(8):
        g.replay()
    torch.cuda.synchronize()
    last = eng.loss_values()[0]
    assert np.isfinite(last) and last < first
    assert int(eng.global_step.item()) == 9        # 1 eager step + 8 replays (capture itself does not execute)


# ---- operator-level checks against torch (fp64 on the same device) ----------------------------------------
def _conv_ref(x, w, b, T):
    N = x.shape[0] // T
    y = torch.nn.functional.conv1d(x.view(N, T, -1).transpose(1, 2), w.permute(2, 1, 0), b, padding='same')
    return y.transpose(1, 2).reshape(x.shape[0], -1)


@pytest.mark.parametrize('M,T,cin,cout,kw', [(96, 12, 128, 128, 3), (4096, 128, 256, 1024, 1), (330, 33, 80, 256, 3),
                                             (2560, 640, 1024, 256, 3), (77, 77, 256, 1025, 1), (64, 16, 336, 256, 2),
                                             # ragged everything: K tails of the 16- and 32-deep tiles, N / M tails, T < tile
                                             (200, 50, 20, 36, 3), (130, 65, 44, 100, 5), (35, 5, 12, 8, 4),
                                             (2048, 128, 1024, 128, 3),        # few tiles, long reduction: split-K forward + bias/act pass
                                             # shapes that select the 128x128x16 tile configurations (fwd / dX / dW)
                                             (8192, 8192, 64, 4096, 1), (8192, 8192, 4096, 64, 1), (256, 256, 4096, 2048, 1)])
def test_conv_gemm_forward_backward(M, T, cin, cout, kw):
    from tacotron_multispeaker_amd._lib import lib, stream
    torch.manual_seed(M + kw)
    dev = 'cuda'
    ldw = (cout + 3) & ~3
    x = torch.randn(M, cin, device=dev)
    w = torch.zeros(kw, cin, ldw, device=dev); w[:, :, :cout] = torch.randn(kw, cin, cout, device=dev) / np.sqrt(cin * kw)
    b = torch.randn(ldw, device=dev)
    y = torch.empty(M, cout, device=dev)
    lib.taco_conv_gemm_fwd(x, w, b, y, M, T, cin, cout, kw, 0, cin, ldw, cout, 1, 0, stream())
    x64 = x.double().requires_grad_(True)
    w64 = w[:, :, :cout].double().requires_grad_(True)
    ref = torch.relu(_conv_ref(x64, w64, b[:cout].double(), T))
    assert rel(y.cpu().numpy(), ref.detach().cpu().numpy()) < 1e-5
    dy = torch.zeros(M, ldw, device=dev); dy[:, :cout] = torch.randn(M, cout, device=dev)
    pre = _conv_ref(x64, w64, b[:cout].double(), T)
    pre.backward(dy[:, :cout].double())
    dx = torch.empty(M, cin, device=dev)
    lib.taco_conv_gemm_bwd_data(dy, w, dx, M, T, cin, ldw, kw, 0, ldw, ldw, cin, 0, stream())
    assert rel(dx.cpu().numpy(), x64.grad.cpu().numpy()) < 1e-5
    dw = torch.zeros(kw, cin, ldw, device=dev)
    lib.taco_conv_gemm_bwd_weight(x, dy, dw, M, T, cin, cout, kw, 0, cin, ldw, ldw, stream())
    assert rel(dw[:, :, :cout].cpu().numpy(), w64.grad.cpu().numpy()) < 1e-5
    assert float(dw[:, :, cout:].abs().max()) == 0 if ldw > cout else True


@pytest.mark.parametrize('M,T,cin,cout,kw,bank', [(13101, 397, 132, 260, 3, 0),      # ragged M / N, K tails (260 % 32, 132 % 32), sequence ends under taps
                                                  (10240, 10240, 256, 1025, 1, 0),     # the linear layer: 1025 columns in rows of 1028
                                                  (20480, 640, 80, 1024, 8, 8),        # post-net bank: 80 channels (flat weight-gradient rows)
                                                  (4096, 128, 128, 2048, 16, 16),      # encoder bank: split over the taps (input gradient)
                                                  (8320, 208, 1024, 256, 3, 0)])
def test_bf16x3_gemms_against_float64(M, T, cin, cout, kw, bank, monkeypatch):
    """The three-MFMA bf16 product kernels (conv_gemm_nn3 / nt3 / tn3; TACO_X3, csrc/gemm.hip) on shapes that reach their tile thresholds:
    forward (mode 2 only), input gradient and weight gradient (default mode 1) against float64.  Tolerance 2e-5 of the result's largest
    element (measured 3-5e-6; exact fp32 products: 5e-7 -- the x3 results must DIFFER from those, which shows the x3 kernels ran)."""
    from tacotron_multispeaker_amd._lib import lib, stream
    dev = 'cuda'
    torch.manual_seed(M + kw)
    ldw = 128 if bank else (cout + 3) & ~3
    taps = kw * (kw + 1) // 2 if bank else kw
    co = 128 if bank else cout
    x = torch.randn(M, cin, device=dev)
    w = torch.zeros(taps, cin, ldw, device=dev); w[:, :, :co] = torch.randn(taps, cin, co, device=dev) / np.sqrt(cin * kw)
    b = torch.randn(max(ldw, cout), device=dev)
    lddy = cout if bank else (cout + 3) & ~3
    dy = torch.zeros(M, lddy, device=dev); dy[:, :cout] = torch.randn(M, cout, device=dev)
    x64 = x.double().requires_grad_(True)
    widths = list(range(1, kw + 1)) if bank else [kw]
    w64, k0 = [], 0
    for k in widths:
        w64.append(w[k0:k0 + k, :, :co].double().requires_grad_(True)); k0 += k
    pre = torch.cat([_conv_ref(x64, wk, b[i * co:(i + 1) * co].double(), T) for i, wk in enumerate(w64)], 1)
    pre.backward(dy[:, :cout].double())
    ref_y = torch.relu(pre).detach().cpu().numpy()
    ref_dx = x64.grad.cpu().numpy()
    ref_dw = torch.cat([wk.grad for wk in w64], 0).cpu().numpy()

    def run(mode):
        monkeypatch.setenv('TACO_X3', mode)
        y = torch.empty(M, cout, device=dev); dx = torch.empty(M, cin, device=dev); dw = torch.zeros_like(w)
        lib.taco_conv_gemm_fwd(x, w, b, y, M, T, cin, cout, kw, bank, cin, ldw, cout, 1, 0, stream())
        lib.taco_conv_gemm_bwd_data(dy, w, dx, M, T, cin, cout if bank else lddy, kw, bank, lddy, ldw, cin, 0, stream())
        lib.taco_conv_gemm_bwd_weight(x, dy, dw, M, T, cin, cout, kw, bank, cin, lddy, ldw, stream())
        torch.cuda.synchronize()
        return y.cpu().numpy(), dx.cpu().numpy(), dw[:, :, :co].cpu().numpy(), dw
    y0, dx0, dw0, _ = run('0')
    y1, dx1, dw1, dwfull = run('1')
    y2, dx2, dw2, _ = run('2')
    assert rel(y0, ref_y) < 2e-6 and rel(dx0, ref_dx) < 3e-6 and rel(dw0, ref_dw) < 3e-6
    assert np.array_equal(y1, y0)                                        # mode 1 leaves the forward pass alone
    for got, ref in ((dx1, ref_dx), (dw1, ref_dw), (y2, ref_y), (dx2, ref_dx), (dw2, ref_dw)):
        assert rel(got, ref) < 2e-5
    assert not np.array_equal(dx1, dx0) and not np.array_equal(dw1, dw0) and not np.array_equal(y2, y0)
    assert float(dwfull[:, :, co:].abs().max()) == 0 if ldw > co else True


@pytest.mark.parametrize('N,T,cin,K', [(3, 20, 128, 16), (2, 35, 80, 8), (32, 128, 128, 16),
                                       # weight gradient over the flattened (tap, channel) rows (Cin not a multiple of the 64-row tile: a tile
                                       # holds channels of two or more taps, each with its own row shift): post-net shape, sequences shorter
                                       # than a 32-row reduction tile, Cin < 64 (three taps in a tile), several reduction splits
                                       (4, 100, 80, 16), (9, 24, 40, 5), (2, 70, 96, 4), (16, 160, 80, 16)])
def test_conv_bank(N, T, cin, K):
    from tacotron_multispeaker_amd._lib import lib, stream
    dev, M, C = 'cuda', N * T, K * 128
    torch.manual_seed(K)
    x = torch.randn(M, cin, device=dev)
    ws = [torch.randn(k, cin, 128, device=dev) / np.sqrt(cin * k) for k in range(1, K + 1)]
    wp = torch.cat(ws, 0).contiguous()
    b = torch.randn(C, device=dev)
    y = torch.empty(M, C, device=dev)
    lib.taco_conv_gemm_fwd(x, wp, b, y, M, T, cin, C, K, K, cin, 128, C, 1, 0, stream())
    x64 = x.double().requires_grad_(True)
    w64 = [w.double().requires_grad_(True) for w in ws]
    pre = torch.cat([_conv_ref(x64, w, b[(k) * 128:(k + 1) * 128].double(), T) for k, w in enumerate(w64)], 1)
    assert rel(y.cpu().numpy(), torch.relu(pre).detach().cpu().numpy()) < 1e-5
    dy = torch.randn(M, C, device=dev)
    pre.backward(dy.double())
    dx = torch.empty(M, cin, device=dev)
    lib.taco_conv_gemm_bwd_data(dy, wp, dx, M, T, cin, C, K, K, C, 128, cin, 0, stream())
    assert rel(dx.cpu().numpy(), x64.grad.cpu().numpy()) < 1e-5
    dw = torch.zeros_like(wp)
    lib.taco_conv_gemm_bwd_weight(x, dy, dw, M, T, cin, C, K, K, cin, C, 128, stream())
    assert rel(dw.cpu().numpy(), torch.cat([w.grad for w in w64], 0).cpu().numpy()) < 1e-5


@pytest.mark.parametrize('N,T,cin,K', [(3, 50, 80, 8), (32, 96, 80, 8)])
def test_conv_bank_by_frame_ranges_and_epilogue_bn_sums(N, T, cin, K):
    """The pieces the post-net bank is computed in behind the decoder pipeline (taco_conv_rows_fwd / taco_conv_rows_bwd_data over frame
    ranges of every sequence) against the whole-tensor launches: forward bit for bit, input gradient to accumulation-order noise (the
    pieces are split over the taps with atomic adds), batch-norm sums from the GEMM epilogues against float64 sums."""
    from tacotron_multispeaker_amd._lib import lib, stream
    dev, M, C = 'cuda', N * T, K * 128
    torch.manual_seed(K + N)
    x = torch.randn(M, cin, device=dev)
    wp = torch.randn(K * (K + 1) // 2, cin, 128, device=dev) / np.sqrt(cin * 4)
    b = torch.randn(C, device=dev)
    y_full, y_rows = torch.empty(M, C, device=dev), torch.full((M, C), 7.0, device=dev)
    st_full, st_rows = torch.zeros(24 * C, dtype=torch.float64, device=dev), torch.zeros(24 * C, dtype=torch.float64, device=dev)
    lib.taco_conv_gemm_bn_fwd(x, wp, b, y_full, M, T, cin, C, K, K, cin, 128, C, 1, st_full, stream())
    cuts = [0, T // 3 - 4, 2 * T // 3 - 4, T]                      # like the engine: a piece ends K // 2 frames short of its chunk
    for t0, t1 in zip(cuts[:-1], cuts[1:]):
        lib.taco_conv_rows_fwd(x, wp, b, y_rows, N, T, t0, t1, cin, C, K, K, cin, 128, C, 1, st_rows, stream())
    torch.cuda.synchronize()
    assert torch.equal(y_full, y_rows)
    y64 = y_full.double()
    for st in (st_full, st_rows):
        sums = st.view(8, 3, C).sum(0)
        assert rel(sums[0].cpu().numpy(), y64.sum(0).cpu().numpy()) < 1e-6
        assert rel(sums[1].cpu().numpy(), (y64 * y64).sum(0).cpu().numpy()) < 1e-6
    dy = torch.randn(M, C, device=dev)
    dx_full = torch.empty(M, cin, device=dev)
    lib.taco_conv_gemm_bwd_data(dy, wp, dx_full, M, T, cin, C, K, K, C, 128, cin, 0, stream())
    init = torch.randn(M, cin, device=dev)
    dx_rows = init.clone()
    for t0, t1 in reversed(list(zip(cuts[:-1], cuts[1:]))):
        lib.taco_conv_rows_bwd_data(dy, wp, dx_rows, N, T, t0, t1, cin, C, K, K, C, 128, cin, 1, stream())
    torch.cuda.synchronize()
    assert rel((dx_rows - init).cpu().numpy(), dx_full.cpu().numpy()) < 1e-5


def test_grouped_weight_and_bias_gradients():
    """taco_wgrad_group / taco_col_sum_group: many independent weight-gradient problems (dense, conv k=3, conv bank, shifted
    recurrent-weight form, ragged shapes) in ONE launch against float64 references; more problems than one group holds."""
    import ctypes
    from tacotron_multispeaker_amd._lib import lib, stream
    from tacotron_multispeaker_amd.engine import TacoWgrad, TacoColSum
    dev = 'cuda'
    torch.manual_seed(11)
    probs = []          # (X, dY, dW, M, T, Cin, Cout, kw, bank, shift, reference)
    def add(M, T, cin, cout, kw=1, bank=0, shift=0):
        C = bank * 128 if bank else cout
        x = torch.randn(M, cin, device=dev)
        dy = torch.randn(M, C, device=dev)
        taps = bank * (bank + 1) // 2 if bank else kw
        dw = torch.zeros(taps, cin, 128 if bank else cout, device=dev)
        x64, dy64 = x.double(), dy.double()
        N = M // T
        xs = x64.view(N, T, cin)
        ref = torch.zeros_like(dw, dtype=torch.float64)
        widths = range(1, bank + 1) if bank else [kw]
        ti = 0
        for k in widths:
            d = dy64.view(N, T, -1)[:, :, (k - 1) * 128:k * 128] if bank else dy64.view(N, T, cout)
            for j in range(k):
                sh = j - (k - 1) // 2 + shift
                xsft = torch.zeros_like(xs)
                if sh >= 0:
                    xsft[:, :T - sh] = xs[:, sh:]
                else:
                    xsft[:, -sh:] = xs[:, :T + sh]
                ref[ti] = torch.einsum('ntc,ntd->cd', xsft, d)
                ti += 1
        probs.append((x, dy, dw, M, T, cin, 128 if bank else cout, 1 if bank else kw, bank, shift, ref))
    add(4096, 4096, 128, 256)
    add(640, 64, 256, 80, kw=3)
    add(384, 48, 128, 0, bank=16)
    add(512, 64, 256, 512, shift=-1)
    add(330, 33, 20, 36, kw=5)
    add(2048, 128, 1024, 128, kw=3)
    for i in range(30):
        add(256 + 64 * (i % 5), 32, 64 + 4 * i, 128 + 4 * (i % 7))
    arr = (TacoWgrad * len(probs))()
    dbs = []            # bias gradients riding on the weight-gradient GEMMs (every second problem asks for one, also conv / bank ones,
                        # whose sums the library takes in a launch of their own)
    for k, (a, (x, dy, dw, M, T, cin, cout, kw, bank, shift, _)) in enumerate(zip(arr, probs)):
        a.X, a.dY, a.dW = x.data_ptr(), dy.data_ptr(), dw.data_ptr()
        a.M, a.T, a.Cin, a.Cout, a.kw, a.bank_K, a.ldx, a.lddy, a.ldw, a.shift = M, T, cin, cout, kw, bank, cin, dy.shape[1], cout, shift
        db = torch.zeros(dy.shape[1], device=dev) if (k % 2 == 0 and not bank) else None
        dbs.append(db)
        a.dbias = db.data_ptr() if db is not None else None
    lib.taco_wgrad_group(ctypes.addressof(arr), len(probs), stream())
    torch.cuda.synchronize()
    for db, (x, dy, dw, M, T, cin, cout, kw, bank, shift, ref) in zip(dbs, probs):
        assert rel(dw.cpu().numpy(), ref.cpu().numpy()) < 1e-5, (M, T, cin, cout, kw, bank, shift)
        if db is not None:
            assert rel(db.cpu().numpy(), dy.double().sum(0).cpu().numpy()) < 1e-5, (M, T, cin, cout, kw, shift)
    cs = (TacoColSum * len(probs))()
    outs = []
    for a, (x, dy, *_r) in zip(cs, probs):
        o = torch.zeros(dy.shape[1], device=dev)
        outs.append(o)
        a.x, a.out, a.ldx, a.M, a.C = dy.data_ptr(), o.data_ptr(), dy.shape[1], dy.shape[0], dy.shape[1]
    lib.taco_col_sum_group(ctypes.addressof(cs), len(probs), stream())
    torch.cuda.synchronize()
    for o, (x, dy, *_r) in zip(outs, probs):
        assert rel(o.cpu().numpy(), dy.double().sum(0).cpu().numpy()) < 1e-5


@pytest.mark.parametrize('M', [64, 200, 4096, 20480 + 17])
def test_fused_highway_stack(M):
    """taco_highway4_fwd / taco_highway4_bwd (csrc/highway.hip): four highway layers (reference models/modules.py:77-90) in one
    launch per direction against a float64 torch composition: layer outputs, saved gate activations, the pre-activation
    gradients dZ of every layer and the input gradient; ragged row counts."""
    import ctypes
    from tacotron_multispeaker_amd._lib import lib, stream
    dev = 'cuda'
    torch.manual_seed(M)
    x0 = torch.randn(M, 128, device=dev)
    W = [torch.randn(128, 256, device=dev) / np.sqrt(128) for _ in range(4)]
    b = [torch.randn(256, device=dev) * 0.3 - torch.cat([torch.zeros(128), torch.ones(128)]).to(dev) for _ in range(4)]
    Z = [torch.empty(M, 256, device=dev) for _ in range(4)]
    y = [torch.empty(M, 128, device=dev) for _ in range(4)]
    pa = lambda ts: (ctypes.c_void_p * 4)(*[t.data_ptr() for t in ts])
    lib.taco_highway4_fwd(x0, pa(W), pa(b), pa(Z), pa(y), M, stream())
    x64 = x0.double().requires_grad_(True)
    W64 = [w.double() for w in W]
    cur, pre, outs = x64, [], []
    for l in range(4):
        z = cur @ W64[l] + b[l].double()
        z.retain_grad()
        pre.append(z)
        # the ReLU on/off decision is taken from the kernel's own saved activation: a pre-activation within fp32 rounding of 0
        # (about one unit in 1e7: expected at the largest M) may fall on either side, and the gradients below are compared on
        # the kernel's linear piece; every such unit must be a float64 near-zero
        on = Z[l][:, :128].double() > 0
        flips = on != (z[:, :128].detach() > 0)
        assert int(flips.sum()) <= 4 and (not bool(flips.any()) or float(z[:, :128].detach()[flips].abs().max()) < 1e-5)
        H, T = z[:, :128] * on, torch.sigmoid(z[:, 128:])
        cur = H * T + cur * (1 - T)
        outs.append(cur)
        assert rel(y[l].cpu().numpy(), cur.detach().cpu().numpy()) < 1e-5
        assert rel(Z[l][:, :128].cpu().numpy(), H.detach().cpu().numpy()) < 1e-5
        assert rel(Z[l][:, 128:].cpu().numpy(), T.detach().cpu().numpy()) < 1e-5
    dy = torch.randn(M, 128, device=dev)
    cur.backward(dy.double())
    dZ = [torch.empty(M, 256, device=dev) for _ in range(4)]
    dx = torch.empty(M, 128, device=dev)
    xin = [x0] + y[:3]
    lib.taco_highway4_bwd(dy, pa(Z), pa(xin), pa(W), pa(dZ), dx, M, stream())
    torch.cuda.synchronize()
    assert rel(dx.cpu().numpy(), x64.grad.cpu().numpy()) < 1e-5
    for l in range(4):
        assert rel(dZ[l].cpu().numpy(), pre[l].grad.cpu().numpy()) < 1e-5, l


def test_empty_and_invalid_arguments_are_rejected():
    from tacotron_multispeaker_amd._lib import lib, stream
    x = torch.zeros(16, 8, device='cuda')
    with pytest.raises(RuntimeError, match='-22'):
        lib.taco_conv_gemm_fwd(x, x, None, x, 0, 1, 8, 8, 1, 0, 8, 8, 8, 0, 0, stream())      # M = 0
    with pytest.raises(RuntimeError, match='-22'):
        lib.taco_conv_gemm_fwd(x, x, None, x, 16, 16, 8, 8, 1, 0, 6, 8, 8, 0, 0, stream())     # ld not a multiple of 4
    with pytest.raises(RuntimeError, match='-22'):
        lib.taco_gru128_seq_fwd(None, 768, x, x, x, x, None, x, 256, x, 1, 1, 2, 0, 1, None, 0, stream())     # null xp
    with pytest.raises(RuntimeError, match='-22'):
        lib.taco_gru128_seq_fwd(x, 768, x, x, x, x, None, x, 256, x, 1, 8, 2, 2, 6, None, 0, stream())        # a chunk without a state buffer
    with pytest.raises(RuntimeError, match='-22'):
        lib.taco_l1_loss_rows(x, 8, x, 8, None, 8, x, 2, 8, 5, 9, 8, 0, 1.0, 0.0, stream())                    # frames beyond T
    with pytest.raises(RuntimeError, match='-22'):
        lib.taco_spin_us(-1, stream())


def test_train_py_end_to_end_with_feeder(tmp_path, monkeypatch):
    """The drop-in flow of the reference's train.py: metadata + .npy files -> DataFeeder thread -> create_model /
    initialize / add_loss / add_optimizer -> step loop with the reference's log line; loss must be finite and fall."""
    import json
    import sys
    rng = np.random.RandomState(0)
    lines = []
    for i in range(12):
        T = 23 + 3 * (i % 5)
        paths = []
        for kind, shape in (('spec', (T, 1025)), ('mel', (T, 80)), ('wav', (T * 250,))):
            p = str(tmp_path / ('%s-%d.npy' % (kind, i)))
            np.save(p, rng.rand(*shape).astype(np.float32))
            paths.append(p)
        lines.append(repr(paths + ['{%s}' % ' '.join('<sym%d>' % rng.randint(0, 7000) for _ in range(4 + i % 7)), i % 3]))
    meta = tmp_path / 'toy_id_num_3.txt'
    meta.write_text('\n'.join(lines) + '\n', encoding='utf-8')
    (tmp_path / 'train_npy_data_dict.json').write_text(json.dumps({'TOY': str(meta)}))
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(sys, 'argv', ['train.py', '--base_dir', str(tmp_path / 'logs'), '--train_data', 'TOY', '--description', 'toy',
                                      '--hparams', 'batch_size=4,outputs_per_step=5,decay_learning_rate=false,initial_learning_rate=0.001',
                                      '--max_steps', '12', '--checkpoint_interval', '5'])
    import importlib
    import hparams as H
    importlib.reload(H)
    import train
    importlib.reload(train)
    train.main()
    log = (tmp_path / 'logs' / 'logs-tacotron-toy' / 'train.log').read_text()
    losses = [float(l.split('loss=')[1].split(',')[0]) for l in log.splitlines() if 'avg_sec/step' in l]
    assert len(losses) >= 12 and all(np.isfinite(losses))
    assert np.mean(losses[-3:]) < np.mean(losses[:3])
    assert 'multi-speaker' in log and 'Saving checkpoint to:' in log
    assert (tmp_path / 'logs' / 'logs-tacotron-toy' / 'model.ckpt-5').exists()
    importlib.reload(H)


@pytest.mark.parametrize('cfg', [(2, 9, 7, 2, 0), (3, 14, 5, 5, 4), (1, 6, 9, 1, 0)])
def test_free_running_inference_matches_oracle(cfg):
    """Synthesis mode (linear_targets=None): batch norm on the moving statistics, last predicted frame fed back
    (reference models/helpers.py:7-38, models/tacotron.py:86-94, synthesizer.py:14-34)."""
    from oracle import tacotron_np as onp
    from tacotron_multispeaker_amd.engine import Engine
    N, Ti, S, r, idn = cfg
    P = onp.init_params(seed=9, r=r, id_num=idn)
    rng = np.random.RandomState(4)
    for k in P:
        if k.endswith('/moving_mean'):
            P[k] = 0.2 * rng.standard_normal(P[k].shape)
        if k.endswith('/moving_variance'):
            P[k] = 0.5 + rng.uniform(0, 1, P[k].shape)
        if k.endswith(('/bias', '/beta')):
            P[k] = P[k] + 0.1 * rng.standard_normal(P[k].shape)
    b = onp.synth_batch(N, Ti, 10 * r, r, seed=5, id_num=idn)
    ref = onp.forward_infer(P, b['inputs'], b['input_lengths'], b['identities'], idn, r, max_iters=S)
    eng = Engine(id_num=idn, r=r, named_params=P)
    i, l, _, _, ids = dev_batch(b, eng.dev)
    mel, lin, al = eng.infer(i, l, ids, steps=S)
    torch.cuda.synchronize()
    assert mel.shape == (N, S * r, 80) and lin.shape == (N, S * r, 1025) and al.shape == (N, Ti, S)
    assert rel(mel.cpu().numpy(), ref['mel_outputs']) < TOL
    assert rel(lin.cpu().numpy(), ref['linear_outputs']) < TOL
    assert rel(al.cpu().numpy(), ref['alignments']) < TOL


def test_free_running_inference_stop_token():
    """models/helpers.py:32-38: a row is finished once a whole r-frame output is exactly zero; decoding ends when all rows
    are.  A zero output projection makes every output exactly 0 at step 0, so both the oracle and the HIP path stop after ONE
    step although max_iters = 7; with a non-zero bias nothing stops and all 7 steps run."""
    from oracle import tacotron_np as onp
    from tacotron_multispeaker_amd.engine import Engine
    N, Ti, r = 3, 11, 5
    P = onp.init_params(seed=9, r=r)
    b = onp.synth_batch(N, Ti, 10 * r, r, seed=5)
    P['output_projection/kernel'] = np.zeros_like(P['output_projection/kernel'])
    P['output_projection/bias'] = np.zeros_like(P['output_projection/bias'])
    ref = onp.forward_infer(P, b['inputs'], b['input_lengths'], None, 0, r, max_iters=7)
    assert ref['mel_outputs'].shape == (N, r, 80)
    eng = Engine(r=r, named_params=P)
    i, l, _, _, _ = dev_batch(b, eng.dev)
    mel, lin, al = eng.infer(i, l, None, max_iters=7)
    torch.cuda.synchronize()
    assert mel.shape == (N, r, 80) and lin.shape == (N, r, 1025) and al.shape == (N, Ti, 1)
    assert float(mel.abs().max()) == 0.0
    assert rel(lin.cpu().numpy(), ref['linear_outputs']) < TOL
    assert rel(al.cpu().numpy(), ref['alignments']) < TOL
    P['output_projection/bias'] = P['output_projection/bias'] + 0.25
    eng2 = Engine(r=r, named_params=P)
    mel2, _, al2 = eng2.infer(i, l, None, max_iters=7)
    ref2 = onp.forward_infer(P, b['inputs'], b['input_lengths'], None, 0, r, max_iters=7)
    assert mel2.shape == (N, 7 * r, 80) == ref2['mel_outputs'].shape and al2.shape == (N, Ti, 7)
    assert rel(mel2.cpu().numpy(), ref2['mel_outputs']) < TOL


def test_checkpoint_resume_reproduces_the_next_step():
    """SURVEY.md 8(f) row f2 (reference train.py:125-129): state_dict at step 5 -> a NEW model -> load_state_dict restores
    parameters, Adam slots, BN moving statistics and global_step bit for bit, and step 6 of the resumed run gives the loss of
    the uninterrupted run.  (Not bit for bit: weight gradients and BN sums accumulate with fp32 / fp64 atomics whose order
    varies from launch to launch; two identical uninterrupted runs differ by the same ~1e-6.)  A checkpoint of another
    layout (outputs_per_step) is rejected."""
    import importlib
    import hparams as H
    importlib.reload(H)
    from models import create_model
    from models.tacotron import GlobalStep
    from oracle import tacotron_np as onp
    H.hparams.parse('outputs_per_step=5,decay_learning_rate=false,initial_learning_rate=0.001')
    bs = [onp.synth_batch(3, 14, 30, 5, seed=60 + i) for i in range(6)]

    def make(seed):
        m = create_model('tacotron', H.hparams)
        m.initialize(bs[0]['inputs'], bs[0]['input_lengths'], bs[0]['mel_targets'], bs[0]['linear_targets'], seed=seed)
        m.add_loss(); m.add_optimizer(GlobalStep())
        return m

    def step(m, b):
        m._set_batch(b['inputs'], b['input_lengths'], b['mel_targets'], b['linear_targets'], None)
        return m.run_step()

    a = make(0)
    for i in range(5):
        out = step(a, bs[i])
    assert out[0] == 5
    sd = a.state_dict()
    torch.save(sd, '/tmp/_taco_ckpt_test.pt')
    loss6 = step(a, bs[5])[1]
    b = make(1)                                   # different initial weights: everything must come from the checkpoint
    b.load_state_dict(torch.load('/tmp/_taco_ckpt_test.pt', weights_only=True))
    os.remove('/tmp/_taco_ckpt_test.pt')
    for k in ('params', 'm', 'v', 'bn'):
        assert torch.equal(getattr(b.engine, k).cpu(), sd[k]), k
    assert int(b.engine.global_step.item()) == 5
    out = step(b, bs[5])
    assert out[0] == 6 and abs(out[1] - loss6) < 1e-5 * abs(loss6)
    H.hparams.parse('outputs_per_step=2')
    c = create_model('tacotron', H.hparams)
    c.initialize(bs[0]['inputs'], bs[0]['input_lengths'], bs[0]['mel_targets'], bs[0]['linear_targets'])
    with pytest.raises(ValueError, match='layout'):
        c.load_state_dict(sd)
    importlib.reload(H)


def test_pipelined_submit_collect_and_snapshot():
    """The loop train.py runs (reference train.py:139-152, pipelined one step deep): submit_step() enqueues a step and its one
    128-byte status copy, collect() waits for that copy only.  With step k+1 submitted before step k is collected the reported
    (global_step, loss, loss_regularity, learning rate, gradient norm) equal those of the synchronous run_step() loop, and a
    snapshot requested with step 2 holds exactly the state after step 2 although steps 3 and 4 were enqueued before it was read
    (what train.py writes as model.ckpt-<step>; checkpoint_id_num recovers the speaker count like synthesizer.py:23-25)."""
    import importlib
    import hparams as H
    importlib.reload(H)
    from models import create_model
    from models.tacotron import GlobalStep, checkpoint_id_num
    from oracle import tacotron_np as onp
    H.hparams.parse('outputs_per_step=5,initial_learning_rate=0.001,decay_learning_rate=false')
    bs = [onp.synth_batch(3, 14 + 2 * i, 30 + 5 * i, 5, seed=80 + i, id_num=4) for i in range(4)]      # a new shape every step

    def make():
        m = create_model('tacotron', H.hparams)
        m.initialize(bs[0]['inputs'], bs[0]['input_lengths'], bs[0]['mel_targets'], bs[0]['linear_targets'],
                     identities=bs[0]['identities'], id_num=4, seed=3)
        m.add_loss(); m.add_optimizer(GlobalStep())
        return m

    def feed(m, b):
        m._set_batch(b['inputs'], b['input_lengths'], b['mel_targets'], b['linear_targets'], b['identities'])

    a = make()
    ref, state2 = [], None
    for i, b in enumerate(bs):
        feed(a, b)
        out = a.run_step()
        ref.append((out[0], out[1], a.mel_loss, a.linear_loss, a.learning_rate, a.max_gradient_norm))
        if i == 1:
            state2 = a.state_dict()
    p = make()
    tickets, got = [], []
    for i, b in enumerate(bs):
        feed(p, b)
        tickets.append(p.submit_step(snapshot=(i == 1)))
        if len(tickets) == 2:                       # one step stays in flight behind the one that is collected
            t = tickets.pop(0)
            out = p.collect(t)
            got.append((out[0], out[1], p.mel_loss, p.linear_loss, p.learning_rate, p.max_gradient_norm, t))
    out = p.collect(tickets[0])
    got.append((out[0], out[1], p.mel_loss, p.linear_loss, p.learning_rate, p.max_gradient_norm, tickets[0]))
    for si, (r_, g_) in enumerate(zip(ref, got)):
        assert g_[0] == r_[0]
        for x, y in zip(r_[1:], g_[1:6]):
            # atomics reorder fp32 sums from launch to launch (1e-6 on step 1); every further step at lr 1e-3 amplifies the difference
            assert abs(x - y) < (1e-5 if si == 0 else 1e-3) * abs(x)
    snap = got[1][6].state_dict()
    assert int(snap['global_step'].item()) == 2 and checkpoint_id_num(snap) == 4
    assert snap['layout']['tf_scope'] == 'model/inference' and any(n == 'embedding_id' for n, _, _ in snap['layout']['entries'])
    for k in ('params', 'm', 'v', 'bn'):
        assert float((snap[k] - state2[k]).abs().max()) < 1e-5, k
    assert float((snap['params'] - p.engine.params.cpu()).abs().max()) > 1e-4       # two more steps have been applied since
    with pytest.raises(ValueError):
        got[0][6].state_dict()                      # submitted without snapshot=True
    st = got[-1][6].status
    assert st['err'] == 0 and st['xcd_checked_clusters'] > 0 and 0 <= st['xcd_fallback_clusters'] <= st['xcd_checked_clusters']
    # a layout with the same parameter count but other entry offsets / names is refused by its signature
    bad = dict(snap); bad['layout'] = dict(snap['layout'], signature='0' * 64, entries=[['embedding', 4, [7352, 256]]])
    with pytest.raises(ValueError, match='different parameter layout'):
        p.load_state_dict(bad)
    importlib.reload(H)


def _toy_dataset(tmp_path, n=12):
    import json
    rng = np.random.RandomState(0)
    lines = []
    for i in range(n):
        T = 23 + 3 * (i % 5)
        paths = []
        for kind, shape in (('spec', (T, 1025)), ('mel', (T, 80)), ('wav', (T * 250,))):
            p = str(tmp_path / ('%s-%d.npy' % (kind, i)))
            np.save(p, rng.rand(*shape).astype(np.float32))
            paths.append(p)
        lines.append(repr(paths + ['{%s}' % ' '.join('<sym%d>' % rng.randint(0, 7000) for _ in range(4 + i % 7)), i % 3]))
    meta = tmp_path / 'toy_id_num_3.txt'
    meta.write_text('\n'.join(lines) + '\n', encoding='utf-8')
    (tmp_path / 'train_npy_data_dict.json').write_text(json.dumps({'TOY': str(meta)}))


def _run_train(tmp_path, monkeypatch, extra):
    import importlib
    import sys
    import hparams as H
    importlib.reload(H)
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(sys, 'argv', ['train.py', '--base_dir', str(tmp_path / 'logs'), '--train_data', 'TOY', '--description', 'toy',
                                      '--hparams', 'batch_size=4,outputs_per_step=5,decay_learning_rate=false,initial_learning_rate=0.001'] + extra)
    import train
    importlib.reload(train)
    train.main()
    importlib.reload(H)
    return (tmp_path / 'logs' / 'logs-tacotron-toy' / 'train.log').read_text()


def test_train_py_restore_step_and_spike_rollback(tmp_path, monkeypatch):
    """reference train.py:125-129 (--restore_step) and :154-160 (loss spike / NaN -> reload the checkpoint
    int((step-10)/interval)*interval and carry on from there).
    Run 1 writes model.ckpt-5 and stops at step 6.  Run 2 resumes from 5: its first step is 6.  Run 3 resumes from 5 and the
    loss is forced to spike at step 18: the driver logs the recovery, reloads model.ckpt-5 (int((18-10)/5)*5) and the next
    step is 6 again.  Run 4 (fresh log dir) spikes at step 3, before any checkpoint exists: the loader's error ends the run
    (reference: saver.restore raises) instead of training on."""
    _toy_dataset(tmp_path)
    log1 = _run_train(tmp_path, monkeypatch, ['--max_steps', '6', '--checkpoint_interval', '5'])
    assert 'Starting new training run' in log1 and 'Saving checkpoint to:' in log1
    ck = tmp_path / 'logs' / 'logs-tacotron-toy'
    assert (ck / 'model.ckpt-5').exists()
    log2 = _run_train(tmp_path, monkeypatch, ['--restore_step', '5', '--max_steps', '8', '--checkpoint_interval', '5'])
    new = log2[len(log1):]
    assert 'Resuming from checkpoint:' in new and 'model.ckpt-5' in new
    steps = [int(l.split('Step')[1].split('[')[0]) for l in new.splitlines() if 'avg_sec/step' in l]
    assert steps[0] == 6 and steps[-1] == 8

    import models.tacotron as MT
    orig = MT.Tacotron.collect
    fired = []

    def spiky_at(k):
        def collect(self, ticket):        # train.py's loop: submit_step() ... collect(); the spike is seen one step late
            out = orig(self, ticket)
            if out[0] == k and not fired:
                fired.append(k)
                return (out[0], out[1] * 100.0, out[2], out[3])
            return out
        return collect
    monkeypatch.setattr(MT.Tacotron, 'collect', spiky_at(18))
    log3 = _run_train(tmp_path, monkeypatch, ['--restore_step', '5', '--max_steps', '20', '--checkpoint_interval', '5'])
    new = log3[len(log2):]
    assert fired == [18] and 'recover to the previous checkpoint' in new
    steps = [int(l.split('Step')[1].split('[')[0]) for l in new.splitlines() if 'avg_sec/step' in l]
    assert steps[0] == 6 and steps[steps.index(18) + 1] == 6 and len(steps) == 20
    assert 'Exiting due to exception' not in new

    del fired[:]
    monkeypatch.setattr(MT.Tacotron, 'collect', spiky_at(3))
    monkeypatch.setattr('sys.argv', [])
    import importlib
    import sys
    import hparams as H
    importlib.reload(H)
    monkeypatch.setattr(sys, 'argv', ['train.py', '--base_dir', str(tmp_path / 'logs'), '--train_data', 'TOY', '--description', 'toy2',
                                      '--hparams', 'batch_size=4,outputs_per_step=5', '--max_steps', '9', '--checkpoint_interval', '5'])
    import train
    importlib.reload(train)
    train.main()
    importlib.reload(H)
    log4 = (tmp_path / 'logs' / 'logs-tacotron-toy2' / 'train.log').read_text()
    assert fired == [3] and 'recover to the previous checkpoint' in log4 and 'Exiting due to exception' in log4
    steps = [int(l.split('Step')[1].split('[')[0]) for l in log4.splitlines() if 'avg_sec/step' in l]
    assert steps == [1, 2, 3]


def test_data_parallel_exchange_plumbing_single_rank_rccl():
    """The overlapped gradient exchange on the GPU with a real RCCL communicator of ONE rank (the box has one GPU): the engine
    is told world = 2, so backward launches its four buckets out of band on the communication stream (all-reduce over one
    rank = identity), allreduce_grads() joins them and the optimizer applies the 1/world factor inside its kernels.  The
    result must equal the oracle's update with the gradient halved; the exchange order must be the backward order.
    (Two or more ranks: tests/test_dp_gloo.py on CPU; no multi-GPU hardware in this pipeline -- unmeasured.)"""
    import socket
    import torch.distributed as dist
    from oracle import tacotron_np as onp, tacotron_torch as ot
    from tacotron_multispeaker_amd.engine import Engine
    s_ = socket.socket(); s_.bind(('127.0.0.1', 0)); port = s_.getsockname()[1]; s_.close()
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % port, rank=0, world_size=1,
                            device_id=torch.device('cuda', 0))
    try:
        N, Ti, To, r = 4, 24, 60, 5
        P = onp.init_params(seed=3, r=r)
        b = onp.synth_batch(N, Ti, To, r, seed=8)
        ts = ot.TrainState(P, torch.float64, r=r, tf_sparse_norm=False)
        last = ts.forward_backward(b)
        for k in last['grads']:
            last['grads'][k] = last['grads'][k] * 0.5
        info = ts.apply(last)
        eng = Engine(r=r, named_params=P, tf_sparse_norm=False)
        eng.world = 2
        eng.exposed_events = []
        eng.train_step(*dev_batch(b, eng.dev)[:4])
        torch.cuda.synchronize()
        eng.check_errors()
        assert eng._exchange.order == [0, 1, 2, 3] and len(eng.exposed_events) == 1
        assert eng.exposed_events[0][0].elapsed_time(eng.exposed_events[0][1]) >= 0.0
        assert abs(float(eng.info[0].item()) - info['global_norm']) < 1e-4 * info['global_norm']
        pn = eng.export_named('params')
        for k, v in ts.P.items():
            assert np.abs(pn[k] - v.detach().numpy()).max() < 1e-5, k
    finally:
        dist.destroy_process_group()


def test_two_rank_data_parallel_through_the_engine(request):
    """SURVEY.md 8(e) with two REAL ranks on the one GPU of the box (scripts/dp_two_ranks.py, started by tests/conftest.py before
    this process touched the device): both ranks run Engine.train_step with world = 2 on rank-specific batches; backward launches the
    four gradient buckets in order [0, 1, 2, 3] behind the producer events, the step-1 update equals the float64 oracle's update on
    the averaged gradient, replicas stay bit-identical for 3 steps of changing shapes, no hand-off times out.  The collective itself
    is gloo (RCCL does not take two ranks on one device): RCCL over xGMI stays unmeasured until an 8-GPU record exists."""
    rc, out = getattr(request.config, '_dp_two_ranks', (-1, 'the run was not started (conftest.pytest_collection_finish)'))
    assert rc == 0, out[-4000:]
    assert out.count('PASS') == 2 and 'RESULT: OK' in out
    assert out.count('bucket order [0, 1, 2, 3]') == 6
    # and train.py itself with two ranks (toy dataset, 8 steps, checkpoints every 4): the replicas agree before every enqueue and after
    # every collect, rank 0 writes the checkpoints, both ranks log every step and leave in order
    assert 'TRAIN RESULT: OK' in out


def test_changing_batch_shapes_reuse_the_workspace():
    """With the real feeder every batch has another (N, T_in, T_out).  The engine's workspace only grows and a buffer whose shape
    changes is a new view of the SAME storage that is not cleared (Engine.buf), so every shape must give what a fresh engine gives:
    big -> small (stale data behind and inside the views) -> other strides -> big again, pipelined (S >= 8) and unpipelined decoders,
    free-running inference in between."""
    from oracle import tacotron_np as onp
    from tacotron_multispeaker_amd.engine import Engine
    r = 5
    P = onp.init_params(seed=5, r=r)
    shapes = [(6, 40, 80, 11), (3, 17, 25, 12), (5, 33, 60, 13), (2, 40, 45, 14), (6, 40, 80, 11)]

    def fwd_bwd(eng, b):
        i, l, m, lin, ids = dev_batch(b, eng.dev)
        eng.forward(i, l, m, ids)
        eng.loss(lin)
        eng.backward()
        torch.cuda.synchronize()
        eng.check_errors()
        return dict(mel=eng.mel_outputs.cpu().numpy().copy(), lin=eng.linear_outputs.cpu().numpy().copy(),
                    align=eng.alignments.cpu().numpy().copy(), loss=eng.loss_values()[0], grads=eng.export_named('grads'))

    eng = Engine(r=r, named_params=P)
    for k, (N, Ti, To, seed) in enumerate(shapes):
        b = onp.synth_batch(N, Ti, To, r, seed=seed)
        got = fwd_bwd(eng, b)
        if k == 1:      # an inference pass re-views the same buffers with yet other shapes
            i, l, _, _, _ = dev_batch(b, eng.dev)
            eng.infer(i, l, steps=6)
            torch.cuda.synchronize()
            got = fwd_bwd(eng, b)
        ref = fwd_bwd(Engine(r=r, named_params=P), b)
        assert rel(got['mel'], ref['mel']) < 1e-6 and rel(got['lin'], ref['lin']) < 1e-6 and rel(got['align'], ref['align']) < 1e-6, (k, N, Ti, To)
        assert abs(got['loss'] - ref['loss']) < 1e-6 * abs(ref['loss'])
        gmax = max(float(np.abs(g).max()) for g in ref['grads'].values())
        for n, g in ref['grads'].items():       # weight gradients accumulate with fp32 atomics: order noise only
            d = float(np.abs(got['grads'][n].astype(np.float64) - g).max())
            assert d < 2e-5 * float(np.abs(g).max()) + 1e-7 * gmax, (k, n)      # (bias-before-BN gradients are exactly 0 + noise)


def test_model_api_synthesis_mode():
    """create_model('tacotron', hparams).initialize(inputs, input_lengths) -- the call synthesizer.py:26 makes."""
    import importlib
    import hparams as H
    importlib.reload(H)
    from models import create_model
    H.hparams.parse('outputs_per_step=5,max_iters=4')
    m = create_model('tacotron', H.hparams)
    ids = np.array([[5, 9, 33, 1, 0, 0]], dtype=np.int32)
    m.initialize(ids, np.array([4], dtype=np.int32))
    assert tuple(m.mel_outputs.shape) == (1, 20, 80) and tuple(m.linear_outputs.shape) == (1, 20, 1025)
    assert tuple(m.alignments.shape) == (1, 6, 4)
    assert bool(torch.isfinite(m.linear_outputs).all())
    importlib.reload(H)


def test_synthesizer_wrapper_from_a_training_checkpoint(tmp_path, monkeypatch):
    """reference synthesizer.py:13-58: Synthesizer().load(checkpoint) recovers the speaker count from the checkpoint, synthesize()
    decodes free-running on the GPU and turns the linear spectrogram into a wav on the CPU.  The checkpoint is one train.py would
    write (two steps of a 3-speaker model); the decode of the wrapper equals Engine.infer on the same weights."""
    import importlib
    import hparams as H
    importlib.reload(H)
    from models import create_model
    from models.tacotron import GlobalStep
    from oracle import tacotron_np as onp
    H.hparams.parse('outputs_per_step=5,griffin_lim_iters=3')
    b = onp.synth_batch(3, 10, 30, 5, seed=90, id_num=3)
    m = create_model('tacotron', H.hparams)
    m.initialize(b['inputs'], b['input_lengths'], b['mel_targets'], b['linear_targets'], identities=b['identities'], id_num=3, seed=4)
    m.add_loss(); m.add_optimizer(GlobalStep())
    for _ in range(2):
        m.run_step()
    ck = str(tmp_path / 'model.ckpt-2')
    torch.save(m.state_dict(), ck)
    import synthesizer
    importlib.reload(synthesizer)
    syn = synthesizer.Synthesizer()
    syn.load(ck)
    assert syn.id_num == 3
    monkeypatch.setattr(synthesizer.hparams, 'max_iters', 12)          # 12 decoder steps = 60 frames: keeps Griffin-Lim short
    text = '{%s}' % ' '.join('<sym%d>' % k for k in (5, 17, 300, 42, 7))
    wav_path = str(tmp_path / 'out.wav')
    def synth():
        return syn.synthesize(text, 1, path=wav_path, path_align=str(tmp_path / 'align.npy'))
    # synthesize() sets max_iters through hparams at load(); decode length is what infer ran
    out = synth()
    assert out == b'' and os.path.getsize(wav_path) > 44
    from scipy.io import wavfile
    sr, wav = wavfile.read(wav_path)
    assert sr == 20000 and wav.dtype == np.int16 and np.abs(wav).max() > 0
    S = syn.model.engine.linear_outputs.shape[1] // 5
    assert len(wav) == (S * 5 - 1) * 250 and syn.alignment.shape[1] == S
    # same weights, same ids -> the engine of the training model decodes the same spectrogram
    import text as T
    seq = T.text_to_sequence2(text, ['basic_cleaners'])[:-1]
    e = m.engine
    dev = lambda a, dt: torch.tensor(a, device=e.dev, dtype=dt)
    e.infer(dev([seq], torch.int32), dev([len(seq)], torch.int32), dev([1], torch.int32), max_iters=S)
    assert rel(syn.model.engine.linear_outputs.cpu().numpy(), e.linear_outputs.cpu().numpy()) < 1e-5
    importlib.reload(H)
