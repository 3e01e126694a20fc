"""Synthetic LJSpeech-shaped batches in the in-memory format of reference datasets/datafeeder_npy.py:163-194
(inputs int32 [N,T_in] padded with 0 and terminated by EOS=1; targets fp32 in [0,1), exactly 0 on padded
frames, at least one padded frame, T_out a multiple of outputs_per_step).  Recipe: SURVEY.md 8(d)."""
import numpy as np

VOCAB = 7352


def synth_batch(N, Ti, To, r, seed=1234, id_num=0, num_mels=80, num_freq=1025, vocab=VOCAB):
    assert To % r == 0
    rng = np.random.RandomState(seed)
    lens = rng.randint(int(np.ceil(0.6 * Ti)), Ti + 1, size=N)
    lens[rng.randint(N)] = Ti
    inputs = np.zeros((N, Ti), dtype=np.int32)
    for n in range(N):
        inputs[n, :lens[n] - 1] = rng.randint(2, vocab, size=lens[n] - 1)
        inputs[n, lens[n] - 1] = 1
    nfr = rng.randint(int(np.ceil(0.7 * To)), To, size=N)
    nfr[rng.randint(N)] = To - 1
    mel = rng.uniform(0, 1, size=(N, To, num_mels)).astype(np.float32)
    lin = rng.uniform(0, 1, size=(N, To, num_freq)).astype(np.float32)
    for n in range(N):
        mel[n, nfr[n]:] = 0.0
        lin[n, nfr[n]:] = 0.0
    ids = rng.randint(0, id_num, size=N).astype(np.int32) if id_num > 1 else None
    return dict(inputs=inputs, input_lengths=lens.astype(np.int32), mel_targets=mel, linear_targets=lin, identities=ids)


def batch_to_device(b, dev):
    import torch
    order = ('inputs', 'input_lengths', 'mel_targets', 'linear_targets', 'identities')
    return [torch.as_tensor(b[k]).to(dev) if b[k] is not None else None for k in order]
