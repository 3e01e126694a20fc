"""Construction API of the reference model (models/tacotron.py:13-202) on top of the MI355X engine.

Same call sequence as the reference's train.py:101-111:

    model = create_model('tacotron', hparams)
    model.initialize(inputs, input_lengths, mel_targets, linear_targets, identities=..., id_num=...)
    model.add_loss()
    model.add_optimizer(global_step)
    step, loss, _, loss_regularity = model.run_step()        # == sess.run([...]) of train.py:142-146

Arguments of initialize() are either concrete batches (numpy arrays / torch tensors: the step runs eagerly on
them) or the `.inputs ... .identities` handles of datasets.datafeeder_npy.DataFeeder (the TF dequeue tensors of
the reference): then every run_step() dequeues the next batch.  All arithmetic runs in the HIP kernels of
tacotron_multispeaker_amd (no TensorFlow, no torch compute, no CPU fallback).
"""
import numpy as np
import torch

from text.symbols import symbols2
from util.infolog import log
from tacotron_multispeaker_amd.engine import Engine


class GlobalStep(object):
    """Stand-in for `tf.Variable(0, name='global_step')` (train.py:100): the value lives on the device."""

    def __init__(self):
        self.engine = None

    def value(self):
        return 0 if self.engine is None else int(self.engine.global_step.item())


class Tacotron():
    def __init__(self, hparams):
        self._hparams = hparams
        self.engine = None
        self._feeder = None
        self._static = None
        self._global_step = None

    # ------------------------------------------------------------------------------------------------
    def initialize(self, inputs, input_lengths, mel_targets=None, linear_targets=None, identities=None, id_num=0,
                   named_params=None, seed=0, device='cuda'):
        '''Builds the model.  Sets "mel_outputs", "linear_outputs" and "alignments" (reference :18-124).'''
        hp = self._hparams
        is_training = linear_targets is not None
        multi = identities is not None and id_num > 1
        self.is_training = is_training
        self._id_num = id_num if multi else 0
        self.engine = Engine(vocab=len(symbols2), embed_text=hp.embedding_text_channels,
                             embed_id=hp.embedding_id_channels, id_num=self._id_num, r=hp.outputs_per_step,
                             num_mels=hp.num_mels, num_freq=hp.num_freq, sample_rate=hp.sample_rate,
                             init_lr=hp.initial_learning_rate, decay_lr=hp.decay_learning_rate, beta1=hp.adam_beta1,
                             beta2=hp.adam_beta2, device=device, seed=seed, named_params=named_params)
        log('multi-speaker' if multi else 'single speaker')
        feeds = [inputs, input_lengths, mel_targets, linear_targets, identities]
        self._feeder = next((f.feeder for f in feeds if hasattr(f, 'feeder')), None)
        if self._feeder is None:
            self._set_batch(*feeds)
            if is_training or mel_targets is not None:
                self._forward()
            else:      # synthesis (synthesizer.py:14-34): free-running decoder, batch norm on the moving statistics
                e, st = self.engine, self._static
                e.infer(st[0], st[1], st[4], max_iters=hp.max_iters)
                self.mel_outputs, self.linear_outputs, self.alignments = e.mel_outputs, e.linear_outputs, e.alignments
        self.inputs, self.input_lengths = inputs, input_lengths
        self.identities, self.mel_targets, self.linear_targets = identities, mel_targets, linear_targets
        r = hp.outputs_per_step
        log('Initialized Tacotron model. Dimensions: ')
        log('embedding:                 %d' % (hp.embedding_text_channels + (hp.embedding_id_channels if multi else 0)))
        log('prenet out:                %d' % 128)
        log('encoder out:               %d' % 256)
        log('attention out:             %d' % 256)
        log('concat attn & out:         %d' % 512)
        log('decoder cell out:          %d' % 256)
        log('decoder out (%d frames):   %d' % (r, hp.num_mels * r))
        log('decoder out (1 frame):     %d' % hp.num_mels)
        log('postnet out:               %d' % 256)
        log('linear out:                %d' % hp.num_freq)

    def _to_dev(self, x, dtype):
        if x is None:
            return None
        t = torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x)
        return t.to(device=self.engine.dev, dtype=dtype).contiguous()

    # ---- host -> device staging for the feeder path -------------------------------------------
    # A batch is ~91 MB at C2 (linear targets 84 MB): it is copied into PINNED host buffers and sent with an
    # asynchronous copy on a dedicated stream while the previous step is still computing (2-deep), so PCIe time
    # (~1.5 ms at Gen5 x16) is off the step's critical path.
    def _stage(self, batch):
        dev = self.engine.dev
        if not hasattr(self, '_copy_stream'):
            self._copy_stream = torch.cuda.Stream(device=dev)
            self._pinned = [dict(), dict()]
            self._stage_idx = 0
        slot = self._pinned[self._stage_idx]
        self._stage_idx ^= 1
        names = ('inputs', 'input_lengths', 'mel_targets', 'linear_targets', 'wavs', 'identities')
        dtypes = (torch.int32, torch.int32, torch.float32, torch.float32, None, torch.int32)
        out = []
        with torch.cuda.stream(self._copy_stream):
            for name, dt, arr in zip(names, dtypes, batch):
                if dt is None or (name == 'identities' and not self._id_num):
                    out.append(None)
                    continue
                src = torch.as_tensor(np.ascontiguousarray(arr)).to(dt)
                flat = slot.get(name)          # pinned capacity only grows (batch shapes change every step with the real feeder)
                if flat is None or flat.numel() < src.numel():
                    flat = torch.empty(max(src.numel(), 1), dtype=dt).pin_memory()
                    slot[name] = flat
                buf = flat[:src.numel()].view(src.shape)
                buf.copy_(src)
                out.append(buf.to(dev, non_blocking=True))
            ev = torch.cuda.Event()
            ev.record(self._copy_stream)
        return out, ev, batch

    def _next_staged(self):
        """Returns the staged batch for this step and starts staging the following one."""
        if getattr(self, '_staged', None) is None:
            b = self._feeder.dequeue()
            if b is None:
                return None
            self._staged = self._stage(b)
        cur = self._staged
        nxt = self._feeder.dequeue(timeout=0.001) if self._feeder._queue.qsize() > 0 else None
        self._staged = self._stage(nxt) if nxt is not None else None
        return cur

    def _set_batch(self, inputs, input_lengths, mel_targets, linear_targets, identities):
        hp = self._hparams
        if mel_targets is not None and np.shape(mel_targets)[1] // hp.outputs_per_step > hp.max_iters:
            raise ValueError('T_out/outputs_per_step exceeds hparams.max_iters (tacotron.py:92-94)')
        self._static = (self._to_dev(inputs, torch.int32), self._to_dev(input_lengths, torch.int32),
                        self._to_dev(mel_targets, torch.float32), self._to_dev(linear_targets, torch.float32),
                        self._to_dev(identities, torch.int32) if self._id_num else None)

    def _forward(self):
        e, s = self.engine, self._static
        e.forward(s[0], s[1], s[2], s[4], training=self.is_training)
        self.mel_outputs, self.linear_outputs, self.alignments = e.mel_outputs, e.linear_outputs, e.alignments

    # ------------------------------------------------------------------------------------------------
    def add_loss(self):
        '''Adds loss to the model. Sets "loss" field. initialize must have been called (reference :127-171).'''
        hp = self._hparams
        # loss_regularity (reference :140-171): computed on the GPU together with its gradient wrt the alignments
        self.engine.set_regularity(hp.overwrought, hp.oneorder_dynamic, hp.variance_between_row, hp.alignment_entropy)
        self.loss_regularity = 0.0
        self.mel_loss = self.linear_loss = self.loss = None
        if self._feeder is None and self._static[3] is not None:
            self.engine.loss(self._static[3], with_grad=False)
            self.loss, self.mel_loss, self.linear_loss = self.engine.loss_values()
            self.loss_regularity = self.engine.loss_regularity

    def add_optimizer(self, global_step):
        '''Adds optimizer. Sets "gradients" and "optimize" fields (reference :174-195).'''
        self._global_step = global_step
        if isinstance(global_step, GlobalStep):
            global_step.engine = self.engine
        self.gradients = self.engine.grads
        self.optimize = self.run_step
        self.learning_rate = _learning_rate_decay(self._hparams.initial_learning_rate, 0) \
            if self._hparams.decay_learning_rate else self._hparams.initial_learning_rate

    # ------------------------------------------------------------------------------------------------
    def run_step(self):
        """One optimizer step == sess.run([global_step, loss, optimize, loss_regularity]) (train.py:142-146).
        Returns (global_step after the step, loss, None, loss_regularity)."""
        e = self.engine
        if self._feeder is not None:
            staged = self._next_staged()
            if staged is None:
                return None
            dev_t, ev, batch = staged
            torch.cuda.current_stream().wait_event(ev)
            hp = self._hparams
            if dev_t[2].shape[1] // hp.outputs_per_step > hp.max_iters:
                raise ValueError('T_out/outputs_per_step exceeds hparams.max_iters (tacotron.py:92-94)')
            self._static = (dev_t[0], dev_t[1], dev_t[2], dev_t[3], dev_t[5])
            self.last_batch = batch
        s = self._static
        e.train_step(s[0], s[1], s[2], s[3], s[4])
        self.mel_outputs, self.linear_outputs, self.alignments = e.mel_outputs, e.linear_outputs, e.alignments
        self.loss, self.mel_loss, self.linear_loss = e.loss_values()
        self.loss_regularity = e.loss_regularity
        e.check_errors()        # the stream is idle after the loss read-back: a timed-out cluster hand-off must not train on
        step = int(e.global_step.item())
        self.learning_rate = float(e.info[1].item())
        return step, self.loss, None, self.loss_regularity

    # checkpoint in the build's own format (TF checkpoints cannot be read without TF)
    def state_dict(self):
        e = self.engine
        return dict(params=e.params.cpu(), m=e.m.cpu(), v=e.v.cpu(), bn=e.bn.cpu(), global_step=e.global_step.cpu(),
                    layout=dict(id_num=self._id_num, r=e.r))

    def load_state_dict(self, sd):
        e = self.engine
        lay = sd.get('layout', {})
        mine = dict(id_num=self._id_num, r=e.r)
        if {k: int(lay.get(k, -1)) for k in mine} != mine or tuple(sd['params'].shape) != tuple(e.params.shape):
            raise ValueError('checkpoint layout %s (%d parameters) does not match the model %s (%d parameters)'
                             % (lay, sd['params'].numel(), mine, e.params.numel()))
        e.params.copy_(sd['params']); e.m.copy_(sd['m']); e.v.copy_(sd['v']); e.bn.copy_(sd['bn'])
        e.global_step.copy_(sd['global_step'])


def _learning_rate_decay(init_lr, global_step):
    # Noam scheme (reference :198-202); the device-side twin lives in csrc/optim.hip
    warmup_steps = 4000.0
    step = float(global_step + 1)
    return init_lr * warmup_steps ** 0.5 * min(step * warmup_steps ** -1.5, step ** -0.5)
