"""Generates tests/golden/feeder_batch.npz by running the REFERENCE's own pure-NumPy batch assembly
(/root/reference/datasets/datafeeder_npy.py:_prepare_batch and helpers, :163-194) on deterministic examples.
TensorFlow / unidecode / inflect are not installed, so they are replaced by inert module stubs for the import
(the functions used here never touch them).  Run in the build container only; the fixture is
data (inputs + expected outputs), the reference source is not copied."""
import os
import random
import sys
from unittest import mock

import numpy as np

REF = '/root/reference'


def examples():
    rng = np.random.RandomState(42)
    out = []
    for i, (L, T) in enumerate([(5, 13), (7, 21), (3, 9), (6, 20)]):
        ids = rng.randint(2, 7352, size=L).astype(np.int32)
        ids[-1] = 1
        mel = rng.randint(0, 100, size=(T, 80)).astype(np.float32) / 100
        lin = rng.randint(0, 100, size=(T, 1025)).astype(np.float32) / 100
        wav = rng.randint(-50, 50, size=T * 250).astype(np.float32) / 50
        out.append((ids, mel, lin, L, wav, 3 + i))
    return out


def main():
    for m in ('tensorflow', 'unidecode', 'inflect'):
        sys.modules[m] = mock.MagicMock()
    os.chdir(REF)
    sys.path.insert(0, REF)
    from datasets import datafeeder_npy as ref
    res = {}
    for r in (5, 2, 1):
        random.seed(1234)
        o = ref._prepare_batch(examples(), r)
        for name, a in zip(('inputs', 'input_lengths', 'mel_targets', 'linear_targets', 'wavs', 'identities'), o):
            res['r%d_%s' % (r, name)] = a
    res['round_up'] = np.array([[x, m, ref._round_up(x, m)] for x in range(1, 40) for m in (1, 2, 3, 5)])
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'feeder_batch.npz')
    np.savez_compressed(out, **res)
    print('wrote', out, os.path.getsize(out))


if __name__ == '__main__':
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    main()
