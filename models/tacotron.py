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
import queue
import threading

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
        self._stager = None
        self._pending = None
        self._status_ring = None
        self._tickets = 0

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
    # A batch is ~91 MB at C2 (linear targets 84 MB).  A stager thread (_Stager below) takes the feeder's numpy batches, copies
    # them into a ring of PINNED host buffers (the copy releases the GIL) and starts the asynchronous host-to-device copies on a
    # copy stream into a ring of device buffers, up to two batches ahead of the running step: neither the host copy nor the PCIe
    # time (~1.5 ms at Gen5 x16) is on the step's critical path, and no buffer is allocated after the largest shape has been seen.
    def _next_staged(self):
        """(device tensors, ready event, numpy batch, ring slot) of the next batch; None once the feeder has stopped."""
        if self._stager is None:
            self._stager = _Stager(self._feeder, self.engine.dev, bool(self._id_num), self.engine.copy_stream())
            self._stager.start()
        return self._stager.get()

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
        Returns (global_step after the step, loss, None, loss_regularity).  = collect(submit_step()): the host waits ONCE, for
        the one 128-byte status copy of the step."""
        t = self.submit_step()
        return None if t is None else self.collect(t)

    STATUS_RING = 4            # status slots = the most tickets that may be outstanding

    def next_batch_ready(self):
        """True when submit_step() has a batch to run (blocks until the stager has one or the feeder has stopped).  Data-parallel
        drivers call it on every rank and agree on the answers BEFORE any rank enqueues the step and its gradient all-reduce."""
        if self._feeder is None:
            return True
        if self._pending is None:
            self._pending = self._next_staged()
        return self._pending is not None

    def submit_step(self, snapshot=False):
        """Enqueue one training step WITHOUT waiting for it and return a ticket for collect().  A driver that keeps one ticket
        outstanding (train.py: submit step k+1, then collect step k) never lets the GPU run dry while the host enqueues: the
        step's scalars travel in one asynchronous copy behind an event (Engine.status_async).  snapshot=True also clones the
        model state right behind the step (stream-ordered, device to device), so a checkpoint of exactly this step can be written
        after later steps have been submitted (Ticket.state_dict()).  Returns None once the feeder has stopped."""
        e = self.engine
        slot = batch = None
        if self._feeder is not None:
            if not self.next_batch_ready():
                return None
            staged, self._pending = self._pending, None
            dev_t, ev, batch, slot = staged
            torch.cuda.current_stream().wait_event(ev)
            hp = self._hparams
            if dev_t[2].shape[1] // hp.outputs_per_step > hp.max_iters:
                raise ValueError('T_out/outputs_per_step exceeds hparams.max_iters (tacotron.py:92-94)')
            self._static = (dev_t[0], dev_t[1], dev_t[2], dev_t[3], dev_t[5])
            self.last_batch = batch
        s = self._static
        e.train_step(s[0], s[1], s[2], s[3], s[4])
        self.mel_outputs, self.linear_outputs, self.alignments = e.mel_outputs, e.linear_outputs, e.alignments
        if self._status_ring is None:
            self._status_ring = [torch.zeros(16, dtype=torch.float64).pin_memory() for _ in range(self.STATUS_RING)]
        pinned = self._status_ring[self._tickets % self.STATUS_RING]
        self._tickets += 1
        done = e.status_async(pinned)
        snap = self._snapshot() if snapshot else None
        if slot is not None:
            self._stager.release(slot, done)       # the device buffers of this batch may be refilled once the step has run
        return Ticket(self, done, pinned, e.dims, batch, snap)

    def collect(self, ticket):
        """Wait for a submitted step (the only host wait of the step) -> (global_step, loss, None, loss_regularity); also sets
        .loss .mel_loss .linear_loss .loss_regularity .learning_rate .max_gradient_norm like the reference's fetches/summaries.
        Raises if a persistent cluster kernel reported a hand-off timeout (the optimizer skipped that step on the device)."""
        ticket.done.synchronize()
        st = self.engine.status_decode(ticket.pinned, ticket.dims)
        ticket.status = st
        if st['err'] != 0:
            raise RuntimeError('persistent cluster kernel reported a hand-off timeout; results are invalid')
        self.loss, self.mel_loss, self.linear_loss = st['loss'], st['mel_loss'], st['linear_loss']
        self.loss_regularity = st['loss_regularity']
        self.engine.loss_regularity = st['loss_regularity']
        self.learning_rate = st['learning_rate']
        self.max_gradient_norm = st['global_norm']
        self.last_status = st
        return st['global_step'], st['loss'], None, st['loss_regularity']

    def attach_feeder(self, feeder):
        """Switch to another feeder object (anything with DataFeeder's dequeue()); the stager restarts on it."""
        self.stop()
        self._feeder, self._stager, self._pending = feeder, None, None

    def stop(self):
        """ends the stager thread (train.py calls it when the loop ends)"""
        if self._stager is not None:
            self._stager.stop()

    # checkpoint in the build's own format (TF checkpoints cannot be read without TF): the flat buffers
    # plus their NAMED layout -- every entry's name (the reference's variable scopes under ParamLayout.TF_SCOPE, e.g.
    # 'model/inference/embedding_id', the one name the reference itself reads back, synthesizer.py:25), offset and shape -- so a
    # file describes itself and a file written by a different layout is refused.
    def _layout_dict(self):
        L = self.engine.L
        return dict(id_num=self._id_num, r=self.engine.r, signature=L.signature(), tf_scope=L.TF_SCOPE, entries=L.describe())

    def _snapshot(self):
        e = self.engine
        return dict(params=e.params.clone(), m=e.m.clone(), v=e.v.clone(), bn=e.bn.clone(), global_step=e.global_step.clone(),
                    layout=self._layout_dict())

    def state_dict(self):
        return {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in self._snapshot().items()}

    def load_state_dict(self, sd):
        e = self.engine
        lay = sd.get('layout', {})
        mine = dict(id_num=self._id_num, r=e.r)
        if {k: int(lay.get(k, -1)) for k in mine} != mine or tuple(sd['params'].shape) != tuple(e.params.shape):
            raise ValueError('checkpoint layout %s (%d parameters) does not match the model %s (%d parameters)'
                             % ({k: lay.get(k) for k in mine}, sd['params'].numel(), mine, e.params.numel()))
        if lay.get('signature') is not None and lay['signature'] != e.L.signature():
            theirs = {n: (o, tuple(sh)) for n, o, sh in lay.get('entries', [])}
            ours = {n: (o, tuple(sh)) for n, o, sh in e.L.describe()}
            bad = sorted(set(theirs) ^ set(ours)) + sorted(n for n in set(theirs) & set(ours) if theirs[n] != ours[n])
            raise ValueError('checkpoint was written with a different parameter layout (first differing entries: %s)' % bad[:5])
        e.params.copy_(sd['params']); e.m.copy_(sd['m']); e.v.copy_(sd['v']); e.bn.copy_(sd['bn'])
        e.global_step.copy_(sd['global_step'])


def checkpoint_id_num(sd):
    """Speaker count of a checkpoint (the reference's synthesizer.py:23-25 reads the shape of 'model/inference/embedding_id'
    from the TF checkpoint): rows of the 'embedding_id' entry of the named layout, 0 for a single-speaker model."""
    lay = sd.get('layout', {})
    for name, _, shape in lay.get('entries', []):
        if name == 'embedding_id':
            return int(shape[0])
    return int(lay.get('id_num', 0))


class Ticket(object):
    """A submitted, not yet collected training step (Tacotron.submit_step)."""

    def __init__(self, model, done, pinned, dims, batch, snap):
        self.model, self.done, self.pinned, self.dims, self.batch, self.snapshot = model, done, pinned, dims, batch, snap
        self.status = None

    def state_dict(self):
        """Host copy of the state cloned right behind this step (submit_step(snapshot=True))."""
        if self.snapshot is None:
            raise ValueError('this step was submitted without snapshot=True')
        return {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in self.snapshot.items()}


class _Stager(threading.Thread):
    """numpy batches of the feeder -> device, off the training thread.  RING slots of (pinned host buffers, device buffers), both
    grow-only; slot reuse is ordered on the device (the copy stream waits for the `done` event of the step that read the slot)
    and on the host (the previous copy out of the pinned buffers is complete before they are overwritten)."""
    RING = 3
    NAMES = ('inputs', 'input_lengths', 'mel_targets', 'linear_targets', 'wavs', 'identities')
    DTYPES = (torch.int32, torch.int32, torch.float32, torch.float32, None, torch.int32)

    def __init__(self, feeder, dev, with_ids, stream=None):
        super(_Stager, self).__init__()
        self.daemon = True
        self.feeder, self.dev, self.with_ids = feeder, dev, with_ids
        self.stream = stream if stream is not None else torch.cuda.Stream(device=dev)
        self.free = queue.Queue()
        self.staged = queue.Queue()
        self.slots = [dict(index=i, pinned={}, device={}, copied=None) for i in range(self.RING)]
        for sl in self.slots:
            self.free.put((sl, None))
        self._stop_flag = False
        self._pool = None
        self.error = None
        self.host_copy_s = 0.0          # seconds spent copying numpy -> pinned (bench.py reports it per batch)
        self.batches = 0

    def get(self):
        item = self.staged.get()
        if item is None:
            self.staged.put(None)       # stays at the end
            if self.error is not None:
                raise self.error
        return item

    def release(self, slot, done_event):
        self.free.put((slot, done_event))

    def stop(self):
        self._stop_flag = True
        self.free.put((None, None))

    COPY_THREADS = 4
    COPY_SPLIT_BYTES = 8 << 20

    def _host_copy(self, dst, src):
        """numpy -> pinned host memory (dtype converted if needed).  Large arrays (the 84 MB linear targets) are cut into row blocks
        copied by a small private thread pool: np.copyto releases the GIL, a single thread moves ~10 GB/s, i.e. 9 ms per C2 batch --
        longer than the training step.  Deliberately NOT torch's intra-op pool: that one is sized by the host's CPU count (256 on
        the GPU boxes, of which a 1-GPU job owns 16), and its spinning workers starve the training thread's kernel launches."""
        if src.nbytes < self.COPY_SPLIT_BYTES or src.shape[0] < 2:
            np.copyto(dst, src, casting='unsafe')
            return
        if self._pool is None:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(self.COPY_THREADS)
        n = src.shape[0]
        cuts = [n * i // self.COPY_THREADS for i in range(self.COPY_THREADS + 1)]
        list(self._pool.map(lambda k: np.copyto(dst[cuts[k]:cuts[k + 1]], src[cuts[k]:cuts[k + 1]], casting='unsafe'),
                            range(self.COPY_THREADS)))

    @staticmethod
    def _grow(store, name, numel, dt, make):
        flat = store.get(name)
        if flat is None or flat.numel() < numel:
            flat = make(max(numel, 1), dt)
            store[name] = flat
        return flat

    def run(self):
        import time
        try:
            if self.dev.index is not None:
                torch.cuda.set_device(self.dev)
            while not self._stop_flag:
                slot, done = self.free.get()
                if slot is None:
                    break
                batch = self.feeder.dequeue()
                if batch is None:
                    break
                if slot['copied'] is not None:
                    slot['copied'].synchronize()        # the last copy OUT of this slot's pinned buffers has finished
                out = []
                with torch.cuda.stream(self.stream):
                    if done is not None:
                        self.stream.wait_event(done)   # the step that read this slot's device buffers has finished
                    t0 = time.perf_counter()
                    for name, dt, arr in zip(self.NAMES, self.DTYPES, batch):
                        if dt is None or (name == 'identities' and not self.with_ids):
                            out.append(None)
                            continue
                        src = np.ascontiguousarray(arr)
                        pin = self._grow(slot['pinned'], name, src.size, dt, lambda n, d: torch.empty(n, dtype=d).pin_memory())
                        dev = self._grow(slot['device'], name, src.size, dt, lambda n, d: torch.empty(n, dtype=d, device=self.dev))
                        pv = pin[:src.size].view(src.shape)
                        self._host_copy(pv.numpy(), src)
                        dv = dev[:src.size].view(src.shape)
                        dv.copy_(pv, non_blocking=True)
                        out.append(dv)
                    self.host_copy_s += time.perf_counter() - t0
                    self.batches += 1
                    ev = torch.cuda.Event()
                    ev.record(self.stream)
                slot['copied'] = ev
                self.staged.put((out, ev, batch, slot))
        except Exception as ex:                         # surfaces in the training thread at its next get()
            self.error = ex
        self.staged.put(None)


def _learning_rate_decay(init_lr, global_step):
    # Noam scheme (reference :198-202); the device-side twin lives in csrc/optim.hip
    warmup_steps = 4000.0
    step = float(global_step + 1)
    return init_lr * warmup_steps ** 0.5 * min(step * warmup_steps ** -1.5, step ** -0.5)
