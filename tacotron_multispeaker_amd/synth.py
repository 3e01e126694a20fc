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


class SyntheticFeeder(object):
    """Stand-in for datasets.datafeeder_npy.DataFeeder with the same consumer interface (`.inputs` ... `.identities` handles
    for Tacotron.initialize, dequeue(), a bounded queue of 8 batches filled by a background thread) that cycles through a pool
    of pre-built synthetic batches instead of reading .npy files: bench.py's train-loop legs measure the step loop of train.py
    (feeder handles -> stager thread -> pinned buffers -> device -> step -> one status read-back) without a dataset on disk.
    What it leaves out is the reference feeder's own cost of producing a batch (np.load, pad, stack: reference
    datasets/datafeeder_npy.py:96-171), which is CPU work of the reference's design and not part of the step."""

    def __init__(self, batches, n_batches=None):
        import queue
        import threading
        from datasets.datafeeder_npy import FeedTensor
        names = ('inputs', 'input_lengths', 'mel_targets', 'linear_targets', 'wavs', 'identities')
        (self.inputs, self.input_lengths, self.mel_targets, self.linear_targets, self.wavs,
         self.identities) = [FeedTensor(self, i, n) for i, n in enumerate(names)]
        self._pool = [(b['inputs'], b['input_lengths'], b['mel_targets'], b['linear_targets'], None, b['identities']) for b in batches]
        self._queue = queue.Queue(maxsize=8)
        self._n = n_batches
        self._stopped = threading.Event()
        self._thread = threading.Thread(target=self._run, daemon=True)

    def start_in_session(self, session=None):
        self._thread.start()

    def stop(self):
        self._stopped.set()

    def _run(self):
        import queue
        i = 0
        while not self._stopped.is_set() and (self._n is None or i < self._n):
            try:
                self._queue.put(self._pool[i % len(self._pool)], timeout=0.2)
                i += 1
            except queue.Full:
                continue
        while not self._stopped.is_set():
            try:
                self._queue.put(None, timeout=0.2)
                return
            except queue.Full:
                continue

    def dequeue(self, timeout=1.0):
        import queue
        while True:
            try:
                return self._queue.get(timeout=timeout)
            except queue.Empty:
                if self._stopped.is_set():
                    return None
