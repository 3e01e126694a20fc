"""CPU tests of the reference-named host surface (hparams, create_model, feeder format, text front end),
the flat parameter layout and the C-ABI library's exported symbols.  No GPU compute."""
import ctypes
import json
import os
import random
import sys

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), 'golden')
sys.path.insert(0, GOLD)


def test_hparams_surface():
    import importlib
    import hparams as H
    importlib.reload(H)
    hp = H.hparams
    v = hp.values()
    assert len(v) == 28 and v['outputs_per_step'] == 1 and v['batch_size'] == 32 and v['sample_rate'] == 20000
    assert v['max_iters'] == 2000 and v['embedding_id_channels'] == 64 and v['initial_learning_rate'] == 0.002
    hp.parse('outputs_per_step=5,batch_size=2,decay_learning_rate=false,initial_learning_rate=0.001,cleaners=basic_cleaners')
    assert hp.outputs_per_step == 5 and hp.batch_size == 2 and hp.decay_learning_rate is False
    assert hp.initial_learning_rate == 0.001 and hp.cleaners == 'basic_cleaners'
    with pytest.raises(ValueError):
        hp.parse('not_a_param=3')
    hp.num_GPU = 2                                    # train.py:55 assigns a new attribute
    assert hp.values()['num_GPU'] == 2
    s = H.hparams_debug_string()
    assert s.startswith('Hyperparameters:\n') and '    outputs_per_step: 5' in s
    importlib.reload(H)


def test_create_model_factory():
    import models
    with pytest.raises(Exception, match='Unknown model: nope'):
        models.create_model('nope', None)
    assert type(models.create_model('tacotron', None)).__name__ == 'Tacotron'


def test_feeder_batch_matches_reference_golden():
    from make_feeder_golden import examples
    from datasets import datafeeder_npy as mine
    g = np.load(os.path.join(GOLD, 'feeder_batch.npz'))
    for r in (5, 2, 1):
        random.seed(1234)
        o = mine._prepare_batch(examples(), r)
        for name, a in zip(('inputs', 'input_lengths', 'mel_targets', 'linear_targets', 'wavs', 'identities'), o):
            ref = g['r%d_%s' % (r, name)]
            assert a.dtype == ref.dtype and a.shape == ref.shape and np.array_equal(a, ref), (r, name)
    for x, m, y in g['round_up']:
        assert mine._round_up(int(x), int(m)) == int(y)


def test_text_front_end_matches_reference_golden(monkeypatch):
    import text
    g = json.load(open(os.path.join(GOLD, 'text_sequences.json')))
    assert g['num_symbols2'] == 7352 == len(text.symbols2)
    monkeypatch.setattr(text, '_symbol_to_id2', dict(g['symbols']))
    id2s = {v: k for k, v in g['symbols'].items()}
    id2s.update({0: '_', 1: '~'})                       # pad / eos head the table (text/symbols.py:23)
    monkeypatch.setattr(text, '_id_to_symbol2', id2s)
    for c in g['cases']:
        assert text.text_to_sequence2(c['text'], ['basic_cleaners']) == c['sequence']
        assert text.sequence_to_text2(c['sequence']) == c['roundtrip']


def test_feeder_thread_end_to_end(tmp_path):
    """metadata file -> background thread -> padded batches (format of datasets/wav_to_npy.py outputs)."""
    import hparams as H
    from datasets.datafeeder_npy import DataFeeder
    from util.coordinator import Coordinator
    hp = H.HParams(**H.hparams.values())
    hp.parse('batch_size=2,outputs_per_step=5')
    rng = np.random.RandomState(0)
    lines = []
    for i in range(6):
        T = 7 + i
        paths = []
        for kind, shape in (('spec', (T, 1025)), ('mel', (T, 80)), ('wav', (T * 250,))):
            p = str(tmp_path / ('%s-%d.npy' % (kind, i)))
            np.save(p, rng.rand(*shape).astype(np.float32))
            paths.append(p)
        lines.append(repr(paths + ['{<sym%d> <sym%d>}' % (i, i + 1), i % 3]))   # placeholder vocab: whole-token symbols
    meta = tmp_path / 'train_id_num_3.txt'
    meta.write_text('\n'.join(lines) + '\n', encoding='utf-8')
    coord = Coordinator()
    random.seed(0)
    feeder = DataFeeder(hp, [str(meta)], coord)
    feeder.start_in_session(None)
    b = feeder.dequeue(timeout=5.0)
    coord.request_stop()
    inputs, lens, mel, lin, wavs, ids = b
    assert inputs.dtype == np.int32 and inputs.shape == (2, 3) and list(lens) == [3, 3] and np.all(inputs[:, -1] == 1)
    assert mel.shape[0] == 2 and mel.shape[2] == 80 and mel.shape[1] % 5 == 0 and lin.shape[1:] == (mel.shape[1], 1025)
    assert np.all(mel[:, -1] == 0) and ids.dtype == np.int32 and wavs.ndim == 2


def test_param_layout_roundtrip_and_counts():
    from tacotron_multispeaker_amd.params import ParamLayout, init_named
    for idn, r, expect in ((0, 5, 8773121), (460, 5, 8818945)):
        L = ParamLayout(id_num=idn, r=r)
        P = init_named(L, seed=1)
        n_train = sum(int(np.prod(v.shape)) for k, v in P.items() if not k.endswith(('moving_mean', 'moving_variance')))
        assert n_train == expect
        flat, bn = torch.zeros(L.total, dtype=torch.float64), torch.zeros(L.bn_total, dtype=torch.float64)
        L.load_named(P, flat, bn)
        back = L.export_named(flat, bn)
        assert set(back) == set(P)
        for k in P:
            assert np.array_equal(back[k], P[k]), k
        assert L.total % 4 == 0 and all(e.offset % 4 == 0 for e in L.entries.values())
        assert L.dense_start == L.entries['prenet/dense_1/kernel'].offset


def test_fresh_initialisers_follow_tf_defaults():
    from tacotron_multispeaker_amd.params import ParamLayout, init_named
    P = init_named(ParamLayout(), seed=0)
    assert np.abs(P['embedding']).max() <= 1.0 and 0.4 < P['embedding'].std() < 0.5      # truncated normal, sigma 0.5
    assert np.all(P['attention_gru/gates/bias'] == 1) and np.all(P['encoder_cbhg/highway_2/T/bias'] == -1)
    k = P['post_cbhg/proj_1/kernel']
    assert np.abs(k).max() <= np.sqrt(6.0 / (3 * 1024 + 3 * 256)) + 1e-12               # glorot uniform with conv fans


def test_c_abi_library_exports_every_declared_symbol():
    from tacotron_multispeaker_amd import _lib, build
    build.build(verbose=False)
    protos = _lib.parse_header()
    assert len(protos) >= 28
    dll = ctypes.CDLL(_lib.LIBPATH)
    for name in protos:
        assert hasattr(dll, name), name


def test_engine_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from tacotron_multispeaker_amd.engine import Engine
    with pytest.raises(RuntimeError, match='no CPU path'):
        Engine()


def test_frame_bands_behind_the_bigru_partition_the_sequence(monkeypatch):
    """Engine._tail_chunks / _new_bands (host logic of the frame-band pipelines): the chunk cuts cover [0, T) in order, and the
    bands reported after each chunk -- frames both directions of the biGRU have passed -- are disjoint and cover every frame
    exactly once, whatever the plan."""
    from tacotron_multispeaker_amd.engine import Engine
    monkeypatch.setenv('TACO_TAIL_MIN_T', '16')
    for plan, Ts in (('0.625:0.775:0.9', (120, 640, 135, 801)), ('0.5:0.7:0.95', (70, 64)), ('0.55:0.8', (135, 33)), ('0.7', (640,))):
        monkeypatch.setenv('TACO_TAIL_PLAN', plan)
        for T in Ts:
            chunks = Engine._tail_chunks(None, T, True)
            assert chunks[0][0] == 0 and chunks[-1][1] == T and all(a[1] == b[0] for a, b in zip(chunks[:-1], chunks[1:]))
            seen = np.zeros(T, dtype=int)
            for (q, p) in chunks:
                for (f0, f1) in Engine._new_bands(T, q, p):
                    assert 0 <= f0 < f1 <= T
                    # complete after step p: the forward direction has done frames < p, the backward direction frames >= T - p
                    assert f1 <= p and f0 >= T - p
                    seen[f0:f1] += 1
            assert (seen == 1).all(), (plan, T, chunks)
    monkeypatch.setenv('TACO_TAIL_PLAN', '0')
    assert Engine._tail_chunks(None, 640, True) == [(0, 640)]          # off by default
    monkeypatch.setenv('TACO_TAIL_PLAN', '0.7')
    assert Engine._tail_chunks(None, 640, False) == [(0, 640)]         # inference never chunks


def test_checkpoint_layout_description_and_signature():
    """The named layout a checkpoint stores (models/tacotron.py: reference synthesizer.py:23-25 recovers id_num from
    model/inference/embedding_id): entries are named, ordered by offset, and the signature tells layouts apart."""
    from tacotron_multispeaker_amd.params import ParamLayout
    from models.tacotron import checkpoint_id_num
    a, b, c = ParamLayout(id_num=0, r=5), ParamLayout(id_num=460, r=5), ParamLayout(id_num=0, r=2)
    assert len({a.signature(), b.signature(), c.signature()}) == 3 and a.signature() == ParamLayout(id_num=0, r=5).signature()
    d = b.describe()
    names = [n for n, _, _ in d]
    assert names[0] == 'embedding' and names[1] == 'embedding_id' and 'bn:post_cbhg/proj_2/moving_variance' in names
    tr = [(n, o, s) for n, o, s in d if not n.startswith('bn:')]
    assert all(x[1] < y[1] for x, y in zip(tr[:-1], tr[1:]))           # flat offsets ascend in creation order
    assert tr[-1][1] + int(np.prod(tr[-1][2])) <= b.total
    assert b.TF_SCOPE == 'model/inference'
    assert checkpoint_id_num({'layout': {'entries': d, 'id_num': 460}}) == 460
    assert checkpoint_id_num({'layout': {'entries': a.describe(), 'id_num': 0}}) == 0


def test_cpu_audio_helpers_round_trip_and_griffin_lim():
    """util/audio.py (synthesis only; north_star keeps Griffin-Lim on the CPU): STFT with the reference's parameters
    (util/audio.py:114-118: n_fft 2048, hop 250, window 1000 at 20 kHz) inverts exactly, Griffin-Lim reduces the spectral error
    of a random-phase start, de-normalisation follows the reference's constants."""
    import copy
    import importlib
    import hparams as H
    importlib.reload(H)
    from util import audio
    importlib.reload(audio)
    hp = H.hparams
    assert audio.stft_parameters(hp) == (2048, 250, 1000)
    t = np.arange(hp.sample_rate // 2) / hp.sample_rate
    y = 0.5 * np.sin(2 * np.pi * 440 * t) + 0.2 * np.sin(2 * np.pi * 1320 * t)
    S = audio.stft(y, hp)
    assert S.shape == (1025, 1 + len(y) // 250)
    yr = audio.istft(S, hp)
    n = min(len(y), len(yr))
    assert np.abs(y[:n] - yr[:n]).max() < 1e-12
    few, more = copy.copy(hp), copy.copy(hp)
    few.griffin_lim_iters, more.griffin_lim_iters = 1, 25
    err = lambda g: np.linalg.norm(np.abs(audio.stft(g, hp))[:, :S.shape[1]] - np.abs(S)) / np.linalg.norm(np.abs(S))
    assert err(audio.griffin_lim(np.abs(S), more)) < err(audio.griffin_lim(np.abs(S), few)) < 1.0
    # a spectrogram that is 1.0 everywhere is ref_level_db above 0 dB: amplitude 10^(20/20) = 10 before the power law
    flat = audio.inv_spectrogram(np.ones((1025, 8)), few)
    assert np.isfinite(flat).all() and len(flat) == 7 * 250
