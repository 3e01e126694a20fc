"""Inference wrapper with the reference's surface (synthesizer.py:13-58): Synthesizer().load(checkpoint) builds the model from
a checkpoint -- the speaker count comes out of the checkpoint itself, like the reference's read of model/inference/embedding_id --
and synthesize(text, identity, path) decodes free-running on the GPU (TacoTestHelper semantics, batch norm on the moving
statistics: Engine.infer), reconstructs the phase on the CPU (util/audio.py) and writes the wav.  Checkpoints are this build's
(`model.ckpt-<step>` written by train.py); TF checkpoints cannot be read without TF."""
import io

import numpy as np
import torch

from hparams import hparams
from models import create_model
from models.tacotron import checkpoint_id_num
from text import sequence_to_text2, text_to_sequence2
from util import audio


class Synthesizer:
    def load(self, checkpoint_path, model_name='tacotron'):
        print('Constructing model: %s' % model_name)
        sd = torch.load(checkpoint_path, weights_only=True)
        self.id_num = checkpoint_id_num(sd)
        hparams.outputs_per_step = int(sd['layout']['r'])       # the output projection's width is part of the checkpoint
        hparams.max_iters = 400                                  # reference :21
        self.model = create_model(model_name, hparams)
        self._sd = sd
        self._loaded = False
        print('Loading checkpoint: %s' % checkpoint_path)

    def synthesize(self, text, identity, path=None, path_align=None):
        cleaner_names = [x.strip() for x in hparams.cleaners.split(',')]
        seq = text_to_sequence2(text, cleaner_names)[:-1]        # reference :39 drops the EOS id
        print(seq)
        print(sequence_to_text2(seq))
        inputs = np.asarray([seq], dtype=np.int32)
        lengths = np.asarray([len(seq)], dtype=np.int32)
        ids = np.asarray([identity], dtype=np.int32)
        m = self.model
        if not self._loaded:
            # the model API decodes at initialize() when no targets are given; the first call also creates the engine, so the
            # weights are put in place and the decode repeated
            m.initialize(inputs, lengths, identities=ids, id_num=self.id_num)
            m.load_state_dict(self._sd)
            self._loaded = True
        e = m.engine
        dev = lambda a: torch.as_tensor(a, device=e.dev)
        e.infer(dev(inputs), dev(lengths), dev(ids) if self.id_num > 1 else None, max_iters=hparams.max_iters)
        linear = e.linear_outputs[0].cpu().numpy()             # [frames, num_freq]
        self.alignment = e.alignments[0].cpu().numpy()
        wav = audio.inv_spectrogram(linear.T, hparams)
        audio.save_wav(wav, path if path is not None else './1.wav', hparams)
        if path_align is not None:
            np.save(path_align, self.alignment)                  # (the reference plots it; matplotlib is not a dependency here)
        out = io.BytesIO()
        return out.getvalue()                                    # the reference returns the (empty) buffer too
