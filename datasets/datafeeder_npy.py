"""Background-thread batch feeder with the on-disk and in-memory formats of the reference
(datasets/datafeeder_npy.py:19-194), rebuilt without TensorFlow queues.

On disk (written by the reference's datasets/wav_to_npy.py): a metadata text file with one Python literal
per line `[spec_path, mel_path, wav_path, text, speaker_id]`; spec .npy float32 [T,1025], mel .npy float32
[T,80], wav .npy float32 [samples].  Speaker ids of the k-th metadata file are offset by the number of
speakers of the files before it (:29-41).

In memory: groups of 20 batches are read, sorted by TEXT length (:105), cut into batches, shuffled (:107);
a batch is (inputs int32 [N,T_in] 0-padded, input_lengths int32 [N], mel_targets fp32 [N,T_out,80],
linear_targets fp32 [N,T_out,1025], wavs fp32 [N,samples], identities int32 [N]) with
T_out = round_up(max_frames + 1, outputs_per_step) (:179-181).  The TF FIFOQueue(8) (:58) becomes a
queue.Queue(maxsize=8); `.inputs` ... `.identities` are handles that `Tacotron.initialize` accepts in place
of the dequeued tensors.
"""
import ast
import json
import os
import queue
import random
import threading
import time
import traceback

import numpy as np

from text import text_to_sequence2
from util.infolog import log

_batches_per_group = 20
_pad = 0


class FeedTensor(object):
    """Stands for one component of `queue.dequeue()` (reference :60): resolved when a batch is dequeued."""

    def __init__(self, feeder, index, name):
        self.feeder, self.index, self.name = feeder, index, name

    def __repr__(self):
        return '<FeedTensor %s>' % self.name


class DataFeeder(threading.Thread):
    '''Feeds batches of data into a queue on a background thread.'''

    def __init__(self, hparams, file_list, coordinator):
        super(DataFeeder, self).__init__()
        self.daemon = True
        self._hparams = hparams
        self._cleaner_names = [x.strip() for x in hparams.cleaners.split(',')]
        self._offset = 0
        self._metadata = []
        self._coord = coordinator
        self._p_phone_sub = 0.5

        id_num = 0
        for file in file_list:
            with open(file, encoding='utf-8') as f:
                id_num_crrt = 0
                crrt_metadata = []
                for line in f:
                    if not line.strip():
                        continue
                    line = list(ast.literal_eval(line))
                    if line[4] > id_num_crrt:
                        id_num_crrt = line[4]
                    line[4] = line[4] + id_num
                    crrt_metadata.append(line)
                id_num += id_num_crrt + 1
                self._metadata = self._metadata + crrt_metadata
                log('No. %d of samples from %s' % (len(crrt_metadata), file))
        random.shuffle(self._metadata)

        self._queue = queue.Queue(maxsize=8)
        names = ('inputs', 'input_lengths', 'mel_targets', 'linear_targets', 'wavs', 'identities')
        (self.inputs, self.input_lengths, self.mel_targets, self.linear_targets, self.wavs,
         self.identities) = [FeedTensor(self, i, n) for i, n in enumerate(names)]

        if hparams.per_cen_phone_input:
            char_2_phone_dict_path = './datasets/char_2_phone_dict.json'
            if not os.path.isfile(char_2_phone_dict_path):
                raise Exception('no char_2_phone dict found')
            with open(char_2_phone_dict_path, 'r') as f:
                self._phone_dict = json.load(f)
                log('Loaded characters to phones dict from %s' % char_2_phone_dict_path)
        else:
            self._phone_dict = None

    def start_in_session(self, session=None):
        self._session = session
        self.start()

    def run(self):
        try:
            while not self._coord.should_stop():
                self._enqueue_next_group()
        except Exception as e:
            traceback.print_exc()
            self._coord.request_stop(e)

    def dequeue(self, timeout=1.0):
        """Next batch as a tuple of numpy arrays (blocks; returns None once the coordinator stopped)."""
        while True:
            try:
                return self._queue.get(timeout=timeout)
            except queue.Empty:
                if self._coord.should_stop():
                    return None

    def _enqueue_next_group(self):
        start = time.time()
        n = self._hparams.batch_size
        r = self._hparams.outputs_per_step
        examples = [self._get_next_example() for i in range(n * _batches_per_group)]
        examples.sort(key=lambda x: x[-3])          # text length (the reference's comment says output length)
        batches = [examples[i:i + n] for i in range(0, len(examples), n)]
        random.shuffle(batches)
        log('Generated %d batches of size %d in %.03f sec' % (len(batches), n, time.time() - start))
        for batch in batches:
            item = _prepare_batch(batch, r)
            while not self._coord.should_stop():
                try:
                    self._queue.put(item, timeout=0.5)
                    break
                except queue.Full:
                    continue

    def _get_next_example(self):
        '''Loads a single example (input, mel_target, linear_target, cost) from disk'''
        if self._offset >= len(self._metadata):
            self._offset = 0
            random.shuffle(self._metadata)
        meta = self._metadata[self._offset]
        self._offset += 1
        text = meta[3]
        if self._phone_dict:
            self._p_phone_sub = random.random() - 0.5 + (self._hparams.per_cen_phone_input * 2 - 0.5)
            text2 = ''
            for word in text.split(' '):
                exist_alpha = any(is_alphabet(item) for item in word)
                phone = self._maybe_get_arpabet(word)
                if not text2 and exist_alpha:
                    text2 = text2 + ' '
                text2 += phone
            text = text2
        input_data = np.asarray(text_to_sequence2(text, self._cleaner_names), dtype=np.int32)
        linear_target = np.load(meta[0])
        mel_target = np.load(meta[1])
        wav = np.load(meta[2])
        identity = meta[4]
        return (input_data, mel_target, linear_target, len(input_data), wav, identity)

    def _maybe_get_arpabet(self, word):
        phone = self._phone_dict.get(word)
        phone = ' '.join(phone) if phone is not None else None
        return '{%s}' % phone if phone is not None and random.random() < self._p_phone_sub else word


def _prepare_batch(batch, outputs_per_step):
    """examples (ids, mel, linear, text_len, wav, speaker) -> padded arrays; the in-batch order is shuffled with
    the global `random` module exactly like the reference (:164)."""
    random.shuffle(batch)
    ids, mels, lins, _, wavs, spk = zip(*batch)
    return (_prepare_inputs(ids),
            np.asarray([len(x) for x in ids], dtype=np.int32),
            _prepare_targets(mels, outputs_per_step),
            _prepare_targets(lins, outputs_per_step),
            _prepare_inputs(wavs),
            np.asarray(spk, dtype=np.int32))


def _prepare_inputs(inputs):
    """1-D sequences -> [N, max_len], right-padded with 0 (the pad symbol id)."""
    width = max(len(x) for x in inputs)
    return np.stack([_pad_input(np.asarray(x), width) for x in inputs])


def _prepare_targets(targets, alignment):
    """[T_i, C] frames -> [N, round_up(max T_i + 1, alignment), C]: always at least one padded frame."""
    frames = _round_up(max(len(t) for t in targets) + 1, alignment)
    return np.stack([_pad_target(np.asarray(t), frames) for t in targets])


def _pad_input(x, length):
    out = np.full((length,), _pad, dtype=x.dtype)
    out[:x.shape[0]] = x
    return out


def _pad_target(t, length):
    out = np.full((length, t.shape[1]), _pad, dtype=t.dtype)
    out[:t.shape[0]] = t
    return out


def _round_up(x, multiple):
    return ((x + multiple - 1) // multiple) * multiple


def is_alphabet(uchar):
    return (u'A' <= uchar <= u'Z') or (u'a' <= uchar <= u'z')
