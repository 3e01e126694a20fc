"""Host-side driver of the MI355X Tacotron training step: owns the flat parameter / gradient / Adam buffers
and the activation workspace (torch tensors = device memory only) and sequences the HIP kernels of
libtaco_hip.so (C-ABI, include/taco_hip.h) on the current HIP stream.  No torch compute op is on the path:
forward, backward and the optimizer are hand-written kernels; everything is asynchronous and free of
host<->device synchronisation, so a full step can be captured into a HIP graph (Engine.capture_step).

Reference being replaced: the TF-1 graph built by models/tacotron.py:35-195 and run at train.py:142-146.
"""
import ctypes
import os
from collections import OrderedDict

import numpy as np
import torch

from ._lib import lib
from .params import ParamLayout, init_named

class TacoWgrad(ctypes.Structure):          # include/taco_hip.h
    _fields_ = [('X', ctypes.c_void_p), ('dY', ctypes.c_void_p), ('dW', ctypes.c_void_p)] + \
               [(n, ctypes.c_int) for n in ('M', 'T', 'Cin', 'Cout', 'kw', 'bank_K', 'ldx', 'lddy', 'ldw', 'shift')] + \
               [('dbias', ctypes.c_void_p)]


class TacoColSum(ctypes.Structure):
    _fields_ = [('x', ctypes.c_void_p), ('out', ctypes.c_void_p), ('ldx', ctypes.c_int), ('M', ctypes.c_int), ('C', ctypes.c_int)]


BN_EPS = 1e-3        # tf.layers.batch_normalization defaults
BN_MOMENTUM = 0.99

ACT_NONE, ACT_RELU = 0, 1

# slots of the attention pointer table (enum TacoAttnPtr in include/taco_hip.h)
_AP = ['W1C', 'F1', 'W2', 'B2', 'WX', 'WHG', 'WHC', 'BG', 'WQ', 'V', 'KEYS', 'MEM', 'ZEROS', 'P1', 'P2', 'R', 'U', 'C',
       'RH', 'HC', 'Q', 'ALIGN', 'DHC', 'DXP', 'DP2', 'DP1', 'DQ', 'DKEYS', 'DMEM', 'DVPART', 'DA', 'DHT', 'DHPART',
       'DHCARRY', 'DCTX', 'DCTXCARRY', 'XCHG', 'ERR', 'DE', 'DCTXS', 'DAEXT']
AP = {n: i for i, n in enumerate(_AP)}


_DEVICE_LOCKS = {}


def _claim_device(dev):
    """The persistent cluster kernels need ALL their workgroups co-resident (they spin on each other's granules), which holds
    only while this process has the GPU to itself: kernels of a second process can keep CUs busy so that a partially
    dispatched cluster spins to its bound (err word) -- or, inside a HIP-graph replay, until the queue watchdog aborts the
    process (round-1 record gpurun_out/time14b.log: two copies of one script started 2 s apart on one GPU; the copy that
    was capturing / replaying the full-step graph dumped core, the same script alone never did).  One process per GPU is the
    deployment model anyway (torch.distributed rank = GPU); this advisory lock turns an accidental second process into a
    clear error instead.  TACO_ALLOW_SHARED_GPU=1 skips it."""
    if os.environ.get('TACO_ALLOW_SHARED_GPU', '0') == '1':
        return
    import fcntl
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    pr = torch.cuda.get_device_properties(idx)
    ident = getattr(pr, 'uuid', None) or '%s-%s-%s' % (getattr(pr, 'pci_domain_id', 0), getattr(pr, 'pci_bus_id', idx),
                                                          getattr(pr, 'pci_device_id', 0))
    key = str(ident)
    if key in _DEVICE_LOCKS:
        return
    # a FIXED directory, never $TMPDIR: two processes started with different TMPDIR (scripts/pmc_step.sh exports its own) must
    # still exclude each other; world-writable so that another user's process meets the same lock
    path = os.path.join('/tmp', 'taco_hip_gpu_%s.lock' % key.replace('/', '_'))
    try:
        f = os.fdopen(os.open(path, os.O_RDWR | os.O_CREAT, 0o666), 'a+')
    except PermissionError:       # the file belongs to another user and is not writable: that user's process may hold the device
        f = open(path, 'r')
    try:
        fcntl.flock(f, fcntl.LOCK_EX | fcntl.LOCK_NB)
    except OSError:
        f.close()
        raise RuntimeError('another process already runs the tacotron_multispeaker_amd engine on GPU %s (%s): the persistent '
                           'cluster kernels need the device to themselves; use one process per GPU' % (idx, path))
    _DEVICE_LOCKS[key] = f


class Engine:
    def __init__(self, vocab=7352, embed_text=256, embed_id=64, id_num=0, r=5, num_mels=80, num_freq=1025,
                 sample_rate=20000, init_lr=0.002, decay_lr=True, beta1=0.9, beta2=0.999, tf_sparse_norm=True,
                 device='cuda', seed=0, named_params=None):
        if not torch.cuda.is_available():
            raise RuntimeError('tacotron_multispeaker_amd.Engine needs an MI355X (HIP) device; there is no CPU path')
        lib.load()
        self.dev = torch.device(device)
        _claim_device(self.dev)
        self.L = ParamLayout(vocab, embed_text, embed_id, id_num, r, num_mels, num_freq)
        L = self.L
        self.r, self.nm, self.nf = r, num_mels, num_freq
        self.npri = int(3000 / (sample_rate * 0.5) * num_freq)       # tacotron.py:134
        self.init_lr, self.decay_lr, self.beta1, self.beta2 = init_lr, decay_lr, beta1, beta2
        self.tf_sparse_norm = tf_sparse_norm
        # optional alignment regularisers (tacotron.py:140-171; hparams overwrought / oneorder_dynamic /
        # variance_between_row / alignment_entropy): weights, all 0.0 = off (set_regularity)
        self.regularity = dict(overwrought=0.0, oneorder_dynamic=0.0, variance_between_row=0.0, alignment_entropy=0.0)
        f = dict(dtype=torch.float32, device=self.dev)
        self.params = torch.zeros(L.total, **f)
        # gradients and the per-step reduction scratch (doubles: BN sums, loss sums, norms) share ONE allocation that a single
        # memset node zero-fills at the start of a step (forward)
        self._zero_doubles = 1 << 18
        self._zbuf = torch.zeros(L.total * 4 + self._zero_doubles * 8, dtype=torch.uint8, device=self.dev)
        self.grads = self._zbuf[:L.total * 4].view(torch.float32)
        self.m = torch.zeros(L.total, **f)
        self.v = torch.zeros(L.total, **f)
        self.bn = torch.zeros(max(L.bn_total, 4), **f)              # moving mean / variance
        self.bnbatch = torch.zeros(max(L.bn_total, 4), **f)         # batch mean / variance of the last step
        self.global_step = torch.zeros(1, dtype=torch.int32, device=self.dev)
        self.dscratch = self._zbuf[L.total * 4:].view(torch.float64)                 # zeroed once per step
        self.info = torch.zeros(4, **f)
        # device status words of the persistent cluster kernels (include/taco_hip.h TACO_AP_ERR): [0] hand-off timeout,
        # [1] clusters that ran on the agent-scope granule fallback although the L2-local form was allowed, [2] clusters checked
        self.err = torch.zeros(4, dtype=torch.int32, device=self.dev)
        self._status_dev = torch.zeros(16, dtype=torch.float64, device=self.dev)
        self._bufs = {}
        self._flat = {}                # name -> flat storage (capacity only grows, see buf())
        self._dpos = 0
        # Streams.  A step needs FOUR concurrent command streams: the caller's (main), two more for the stages of the decoder chunk
        # pipeline (stream_b, stream_c) and one auxiliary stream that carries, at different times of the step, the post-net conv-bank
        # pieces behind the decoder pipeline and the deferred weight-gradient GEMMs (side_streams[0] == stream_d).  Four is also what
        # the hardware offers without penalty: ROCm multiplexes all HIP streams of a process onto GPU_MAX_HW_QUEUES = 4 hardware
        # queues (one per compute pipe), and two streams that share a queue run strictly one after the other.  Measured at C2 (round
        # 3, profiles/r03_stream_queues.md): every stream on its own queue 7.43 ms; the auxiliary stream sharing the main stream's
        # queue 8.57 ms; five or more hardware queues (GPU_MAX_HW_QUEUES >= 5: two active queues per pipe) 10.1 ms.  Which pool
        # stream lands on which queue is the runtime's choice, so the engine does not assume it: _pick_streams() measures it once.
        pr = [int(x) for x in os.environ.get('TACO_PRIO', '0:0').split(':')]
        self.main_stream = torch.cuda.Stream(device=self.dev, priority=pr[0]) if pr[0] != 0 else None
        if pr[0] == pr[1]:
            picked = self._pick_streams(3, pr[0])
        else:
            # TACO_PRIO=-1:0: the critical streams in the high-priority pool (the runtime keeps separate hardware queues per priority),
            # the auxiliary stream in the normal one: still four active queues (the idle null stream shares the auxiliary one's pool)
            picked = self._pick_streams(2, pr[0])
            picked.append(self._pick_streams(1, pr[1], against=picked)[0])
        self.stream_b, self.stream_c = picked[0], picked[1]     # decoder pipeline stages (GRU1 / GRU2 or attention)
        self.side_streams = [picked[2]] + [torch.cuda.Stream(device=self.dev, priority=pr[1])
                                           for _ in range(max(1, int(os.environ.get('TACO_SIDE_STREAMS', '1'))) - 1)]
        if os.environ.get('TACO_MERGE_AUX', '1') != '0':
            self.stream_d = self.side_streams[0]               # post-net conv bank pieces behind the decoder pipeline
        else:
            self.stream_d = torch.cuda.Stream(device=self.dev, priority=pr[0])
        self._side_rr = 0
        self._side_active = False
        self._deferred = []
        self.pipe_chunks = int(os.environ.get('TACO_CHUNKS', '4'))
        self.pipe_chunks_bwd = int(os.environ.get('TACO_CHUNKS_BWD', str(self.pipe_chunks)))
        self.last_chunk_frac = float(os.environ.get('TACO_LAST_CHUNK', '0.7'))     # last chunk length / (S / chunks)
        self.overlap_wgrad = os.environ.get('TACO_OVERLAP_WGRAD', '1') != '0'
        self.group_wgrad = os.environ.get('TACO_GROUP_WGRAD', '1') != '0'      # grouped weight / bias gradient launches
        self.fused_highway = os.environ.get('TACO_FUSED_HIGHWAY', '1') != '0'  # four highway layers in one launch per direction
        self.fuse_bias_grad = os.environ.get('TACO_FUSE_BIAS_GRAD', '1') != '0'   # dense bias gradients ride on the weight-gradient GEMMs
        self.enc_flush_sites = tuple(int(x) for x in os.environ.get('TACO_ENC_FLUSH', '3'))
        self.post_pipe = os.environ.get('TACO_POST_PIPE', '1') != '0'          # post-net conv bank chunk by chunk behind the decoder
        self.no_cluster = os.environ.get('TACO_NO_CLUSTER', '0') == '1'     # force the per-step attention kernels (tests)
        self.world = 1                 # data-parallel replicas (set by train.py / bench.py after init_process_group)
        # bucket all-reduces are ordered behind this stream: a FIFTH stream would share a hardware queue with one of the four above,
        # and the only one whose work it may sit behind without stalling the critical path is the auxiliary stream (the gradient
        # buckets are produced there anyway)
        self.comm_stream = self._queue_mate(self.side_streams[0]) if os.environ.get('TACO_COMM_ON_AUX_QUEUE', '1') != '0' \
            else torch.cuda.Stream(device=self.dev)
        self._exchange = None
        self.exposed_events = None     # bench: list of (event, event) pairs around the wait for the gradient exchange
        self.load_named(named_params if named_params is not None else init_named(L, seed))

    # ---- stream selection ----------------------------------------------------------------------------------
    PROBE_US = 300

    def _serialised(self, a, b):
        """True when kernels on streams a and b run one after the other, i.e. the two streams share a hardware queue: two idle
        kernels of PROBE_US microseconds take 2 x PROBE_US instead of 1 x (host clock around a device synchronisation)."""
        import time
        best = 1e9
        for _ in range(2):
            torch.cuda.synchronize(self.dev)
            t0 = time.perf_counter()
            lib.taco_spin_us(self.PROBE_US, a.cuda_stream)
            lib.taco_spin_us(self.PROBE_US, b.cuda_stream)
            a.synchronize(); b.synchronize()
            best = min(best, (time.perf_counter() - t0) * 1e6)
        return best > 1.6 * self.PROBE_US

    def _pick_streams(self, n, priority, against=()):
        """n streams from torch's pool that are pairwise on different hardware queues and not on the queue of the current (main)
        stream nor of the streams in `against`.  Falls back to the first pool streams when the runtime offers fewer queues
        (GPU_MAX_HW_QUEUES < 4) or TACO_STREAM_PROBE=0."""
        main = self.main_stream if self.main_stream is not None else torch.cuda.current_stream(self.dev)
        cands = [torch.cuda.Stream(device=self.dev, priority=priority) for _ in range(int(os.environ.get('TACO_STREAM_CANDIDATES', '12')))]
        self._stream_pool = cands
        if os.environ.get('TACO_STREAM_PROBE', '1') == '0':
            return cands[:n]
        with torch.cuda.device(self.dev):
            lib.taco_spin_us(1, main.cuda_stream)          # module load / first-launch cost out of the measurement
            torch.cuda.synchronize(self.dev)
            chosen = []
            for c in cands:
                if all(not self._serialised(c, o) for o in [main] + list(against) + chosen):
                    chosen.append(c)
                    if len(chosen) == n:
                        break
        self.stream_probe = dict(found=len(chosen), wanted=n)
        for c in cands:                                     # not enough distinct queues: fill up in pool order
            if len(chosen) < n and c not in chosen:
                chosen.append(c)
        return chosen

    def _queue_mate(self, stream, exclude=()):
        """A pool stream that shares `stream`'s hardware queue (or `stream` itself when the probe finds none)."""
        if os.environ.get('TACO_STREAM_PROBE', '1') == '0':
            return torch.cuda.Stream(device=self.dev)
        with torch.cuda.device(self.dev):
            for c in self._stream_pool:
                if c is not stream and c not in (self.stream_b, self.stream_c) and all(c is not x for x in exclude) \
                        and self._serialised(c, stream):
                    return c
        return stream

    def copy_stream(self):
        """Stream for the host-to-device staging copies of the NEXT batch (models/tacotron.py _Stager).  A copy occupies the
        hardware queue of its stream until it has finished (~2-4 ms for a 91 MB C2 batch), so it must not share a queue with the
        main stream or the decoder pipeline stages: TACO_COPY_STREAM = aux (default: a stream on the auxiliary stream's queue, whose
        work has slack), high (a high-priority stream: the runtime keeps separate queues per priority), pool (next pool stream)."""
        mode = os.environ.get('TACO_COPY_STREAM', 'aux')
        if mode == 'high':
            return torch.cuda.Stream(device=self.dev, priority=-1)
        if mode == 'pool':
            return torch.cuda.Stream(device=self.dev)
        mate = self._queue_mate(self.side_streams[0], exclude=(self.comm_stream,))
        return mate if mate is not self.side_streams[0] else torch.cuda.Stream(device=self.dev)

    # ---- parameters ---------------------------------------------------------------------------------------
    def load_named(self, named):
        cpu = torch.zeros(self.L.total, dtype=torch.float32)
        cpubn = torch.zeros(max(self.L.bn_total, 4), dtype=torch.float32)
        self.L.load_named(named, cpu, cpubn)
        self.params.copy_(cpu)
        self.bn.copy_(cpubn)

    def export_named(self, which='params'):
        flat = {'params': self.params, 'grads': self.grads, 'm': self.m, 'v': self.v}[which]
        return self.L.export_named(flat, self.bn if which == 'params' else None)

    def P(self, name):
        return self.L.view(self.params, name)

    def G(self, name):
        return self.L.view(self.grads, name)

    # ---- workspace ------------------------------------------------------------------------------------------
    def buf(self, name, *shape, dtype=torch.float32):
        """Named workspace tensor.  Capacity only grows: with the real feeder T_in / T_out change from batch to batch, and a
        workspace keyed on the exact shape re-allocated and zero-filled ~150 tensors (1 GB at C2 sizes) on the host's critical path
        every step.  A buffer whose shape changes is a new VIEW of the same storage and is NOT cleared: every kernel writes what it
        (or its consumer) reads; the few tensors that must start at zero are zero-filled explicitly where they are used ('zeros' is
        never written).  tests/test_gpu_parity.py::test_changing_batch_shapes_reuse_the_workspace holds this to fresh engines."""
        t = self._bufs.get(name)
        if t is not None and t.dtype == dtype and tuple(t.shape) == tuple(shape):
            return t
        n = 1
        for d in shape:
            n *= int(d)
        flat = self._flat.get(name)
        if flat is None or flat.dtype != dtype or flat.numel() < n or name == 'zeros':
            flat = torch.zeros(max(n, 1), dtype=dtype, device=self.dev)
            self._flat[name] = flat
        t = flat[:n].view(*shape)
        self._bufs[name] = t
        return t

    def dslot(self, n):
        """n doubles of per-step zeroed reduction scratch."""
        n = (n + 1) & ~1
        if self._dpos + n > self.dscratch.numel():
            raise RuntimeError('double scratch exhausted')
        s = self.dscratch[self._dpos:self._dpos + n]
        self._dpos += n
        return s

    def xchg(self, N):
        """granule scratch of the persistent cluster kernels (8-byte {epoch, value} slots)."""
        return self.buf('xchg', ((N + 1) // 2) * 16 * 256, dtype=torch.int64)

    def check_errors(self):
        """raises if a bounded spin of a persistent kernel timed out (synchronises)."""
        if int(self.err[0].item()) != 0:
            raise RuntimeError('persistent cluster kernel reported a hand-off timeout; results are invalid')

    STATUS_FIELDS = ('mel_sum', 'lin_sum', 'pri_sum', 'loss_regularity', 'global_norm', 'learning_rate', 'clip_factor',
                     'global_step', 'err', 'xcd_fallback_clusters', 'xcd_checked_clusters')

    def status_async(self, pinned16):
        """Everything the step loop reads back (reference train.py:142-152 and the summary scalars of :23-36) in ONE 128-byte
        device-to-host copy: taco_step_status packs the loss sums, loss_regularity, global norm, learning rate, global_step and
        the status words into 16 doubles, which are copied into the PINNED host tensor `pinned16` behind the returned event.
        The host never blocks here; decode with status_decode() after event.synchronize()."""
        lib.taco_step_status(self.loss_sums, self.reg_sum, self.info, self.err, self.global_step, self._status_dev, self.st)
        pinned16.copy_(self._status_dev, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return ev

    def status_decode(self, pinned16, dims=None):
        """-> dict(loss, mel_loss, linear_loss, loss_regularity, global_norm, learning_rate, global_step, err, ...) from a
        completed status_async() slot; dims = the (N, Ti, To, S) of the step it belongs to."""
        N, Ti, To, S = dims or self.dims
        v = [float(x) for x in pinned16.tolist()]
        st = dict(zip(self.STATUS_FIELDS, v))
        mel = st['mel_sum'] / (N * To * self.nm)
        lin = 0.5 * st['lin_sum'] / (N * To * self.nf) + 0.5 * st['pri_sum'] / (N * To * self.npri)
        st.update(mel_loss=mel, linear_loss=lin, loss=mel + lin + st['loss_regularity'], global_step=int(st['global_step']),
                  err=int(st['err']), xcd_fallback_clusters=int(st['xcd_fallback_clusters']),
                  xcd_checked_clusters=int(st['xcd_checked_clusters']))
        return st

    @property
    def st(self):
        return torch.cuda.current_stream().cuda_stream

    # ---- optional in-step kernel timing (bench.py): HIP events around every launch of a kernel family, recorded on the stream
    # the kernel is launched on; ktime = None (default) costs nothing
    ktime = None
    sections = None          # dev: list of (name, event) recorded on the main stream at stage boundaries (scripts/dev_sections.py)

    def _mark(self, name):
        if self.sections is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.sections.append((name, e))

    def _timed(self, family, flops, fn):
        if self.ktime is None:
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        self.ktime.append((family, float(flops), e0, e1))

    # ---- op helpers ---------------------------------------------------------------------------------------------
    @staticmethod
    def _gemm_flops(M, Cin, Cout, kw, bank):
        taps = bank * (bank + 1) // 2 if bank else kw
        return 2.0 * M * Cin * (128 if bank else Cout) * taps

    def gemm(self, X, W, bias, Y, M, Cin, Cout, T=None, kw=1, bank=0, ldx=None, ldw=None, ldy=None, act=0, acc=0):
        self._timed('fwd GEMM (conv_gemm_nn2)', self._gemm_flops(M, Cin, Cout, kw, bank),
                    lambda: lib.taco_conv_gemm_fwd(X, W, bias, Y, M, T or M, Cin, Cout, kw, bank,
                                                   ldx or X.stride(-2), ldw or Cout, ldy or Y.stride(-2), act, acc, self.st))

    def conv_bn(self, X, W, bias, Y, M, Cin, Cout, T, kw=1, bank=0, ldw=None, act=0):
        """conv (+ activation) whose output is batch-normalised next: the BN sums come out of the GEMM epilogue.  Returns the sums."""
        dst = self.dslot(24 * Cout)
        self._timed('fwd GEMM (conv_gemm_nn2)', self._gemm_flops(M, Cin, Cout, kw, bank),
                    lambda: lib.taco_conv_gemm_bn_fwd(X, W, bias, Y, M, T, Cin, Cout, kw, bank, X.stride(-2), ldw or Cout, Y.stride(-2),
                                                      act, dst, self.st))
        return dst

    def gemm_dx(self, dY, W, dX, M, Cin, Cout, T=None, kw=1, bank=0, lddy=None, ldw=None, lddx=None, acc=0):
        self._timed('dX GEMM (conv_gemm_nt2)', self._gemm_flops(M, Cin, Cout, kw, bank),
                    lambda: lib.taco_conv_gemm_bwd_data(dY, W, dX, M, T or M, Cin, Cout, kw, bank,
                                                        lddy or dY.stride(-2), ldw or Cout, lddx or dX.stride(-2), acc, self.st))

    # Weight / bias gradients only feed the flat gradient buffer, so during backward they are enqueued on a SIDE
    # stream (forked from / joined to the main stream with events; captured as a parallel branch of the HIP graph)
    # and fill the CUs that the latency-bound persistent recurrence kernels leave idle.
    def _side(self, fn):
        if not self._side_active:
            return self._emit([fn]) if isinstance(fn, tuple) else fn()
        self._deferred.append(fn)          # released by flush_side() when a recurrence kernel has just been launched

    def _emit(self, items):
        """Launch a run of deferred weight-gradient ('dw', ...) / bias-gradient ('cs', ...) descriptors on the current stream:
        ONE grouped launch each (taco_wgrad_group / taco_col_sum_group; TACO_GROUP_WGRAD=0: one launch per problem)."""
        dw = [it for it in items if it[0] == 'dw']
        cs = [it for it in items if it[0] == 'cs']
        if dw:
            arr = (TacoWgrad * len(dw))()
            fl = 0.0
            for a, (_, X, dY, dW, M, T, Cin, Cout, kw, bank, ldx, lddy, ldw, shift, flops, *dbias) in zip(arr, dw):
                a.X, a.dY, a.dW = X.data_ptr(), dY.data_ptr(), dW.data_ptr()
                a.M, a.T, a.Cin, a.Cout, a.kw, a.bank_K, a.ldx, a.lddy, a.ldw, a.shift = M, T, Cin, Cout, kw, bank, ldx, lddy, ldw, shift
                a.dbias = dbias[0].data_ptr() if dbias else None       # bias gradient riding on this GEMM (colsum())
                fl += flops
            if self.group_wgrad:
                self._timed('dW GEMM (conv_gemm_tn2_group)', fl, lambda: lib.taco_wgrad_group(ctypes.addressof(arr), len(dw), self.st))
            else:
                for a in arr:
                    if a.shift:
                        lib.taco_gemm_tn_shift(a.X, a.dY, a.dW, a.M, a.T, a.Cin, a.Cout, a.ldx, a.lddy, a.ldw, a.shift, self.st)
                    else:
                        lib.taco_conv_gemm_bwd_weight(a.X, a.dY, a.dW, a.M, a.T, a.Cin, a.Cout, a.kw, a.bank_K, a.ldx, a.lddy, a.ldw, self.st)
                    if a.dbias:
                        lib.taco_col_sum(a.dY, a.lddy, a.dbias, a.M, a.Cout, self.st)
        if cs:
            arr = (TacoColSum * len(cs))()
            for a, (_, x, ldx, out, M, C) in zip(arr, cs):
                a.x, a.out, a.ldx, a.M, a.C = x.data_ptr(), out.data_ptr(), ldx, M, C
            if self.group_wgrad:
                lib.taco_col_sum_group(ctypes.addressof(arr), len(cs), self.st)
            else:
                for a in arr:
                    lib.taco_col_sum(a.x, a.ldx, a.out, a.M, a.C, self.st)

    def inputs_ready(self):
        """Event on the current stream: everything the deferred weight-gradient descriptors queued so far read has been produced.
        Record it BEFORE launching a recurrence kernel and hand it to flush_side() after the launch: the side stream then starts
        beside the recurrence instead of behind it (round 3: the linear layer's dW ran after the 0.5 ms post-net biGRU BPTT
        although 192 CUs were idle during it, the decoder's dW after the encoder biGRU BPTT)."""
        ev = torch.cuda.Event()
        ev.record()
        return ev

    def flush_side(self, ready=None):
        """Enqueue the pending weight-gradient work on the side stream.  Called right AFTER a persistent recurrence
        kernel was launched on the main stream: that kernel occupies <= 128 CUs for ~1 ms, the GEMMs fill the rest;
        between recurrences the critical-path GEMMs keep the machine to themselves.  ready: inputs_ready() event taken before
        that launch (default: an event recorded now, i.e. the side stream also waits for the kernel just launched)."""
        if not self._deferred:
            return
        ev = ready
        if ev is None or os.environ.get('TACO_FLUSH_BESIDE', '1') == '0':
            ev = torch.cuda.Event()
            ev.record()
        for ss in self.side_streams:
            ss.wait_event(ev)
        # runs of weight / bias gradient descriptors become grouped launches; with several side streams the runs are dealt
        # round-robin
        run = []

        def emit_run():
            if run:
                with torch.cuda.stream(self.side_streams[self._side_rr % len(self.side_streams)]):
                    self._emit(run)
                self._side_rr += 1
                del run[:]
        drop = os.environ.get('TACO_DEV_DROP_WGRAD') == '1'       # developer bound: the step WITHOUT its deferred weight gradients (wrong results)
        for fn in self._deferred:
            if isinstance(fn, tuple) and fn[0] in ('dw', 'cs'):
                if not drop:
                    run.append(fn)
                continue
            emit_run()
            if isinstance(fn, tuple):      # ('bucket', go): every producer of a gradient bucket is now enqueued
                fn[1]()
                continue
            with torch.cuda.stream(self.side_streams[self._side_rr % len(self.side_streams)]):
                fn()
            self._side_rr += 1
        emit_run()
        self._deferred = []

    # ---- data-parallel exchange overlapped with backward (tacotron_multispeaker_amd/dp.py) --------------
    def _bucket_ready(self, k):
        """Called by backward() where the LAST producer of gradient bucket k (dp.bucket_ranges order) has just been created.
        Main-stream producers are captured by an event recorded here; side-stream producers (deferred weight-gradient GEMMs)
        by a marker in the deferred queue: when flush_side() reaches it they are all enqueued, the communication stream is
        made to wait for them and the bucket's all-reduce is launched behind it -- while the BPTT of the earlier layers is
        still running.  Program order is identical on every rank, so the collectives are issued in the same order."""
        if self.world <= 1 or self._exchange is None or k >= len(self._exchange.ranges):
            return
        ev = torch.cuda.Event()
        ev.record()

        def go():
            cs = self.comm_stream
            cs.wait_event(ev)
            for ss in self.side_streams:
                cs.wait_stream(ss)
            with torch.cuda.stream(cs):
                self._exchange.launch(k)
        if self._side_active:
            self._deferred.append(('bucket', go))
        else:
            go()

    def gemm_dw(self, X, dY, dW, M, Cin, Cout, T=None, kw=1, bank=0, ldx=None, lddy=None, ldw=None):
        self._side(('dw', X, dY, dW, M, T or M, Cin, Cout, kw, bank, ldx or X.stride(-2), lddy or dY.stride(-2), ldw or Cout, 0,
                    self._gemm_flops(M, Cin, Cout, kw, bank)))

    def gemm_dw_shift(self, X, dY, dW, M, T, K, N, ldx, lddy, ldw, shift=-1):
        self._side(('dw', X, dY, dW, M, T, K, N, 1, 0, ldx, lddy, ldw, shift, 2.0 * M * K * N))

    def colsum(self, x, out, M, C, ldx=None):
        """out[c] += sum_m x[m, c] (bias gradient).  When the weight-gradient GEMM of the same dense layer is still queued (same dY,
        same rows, kw = 1), the sums ride on it: its workgroups add up the dY tiles they stage in LDS anyway (TacoWgrad.dbias) and
        no pass of its own reads dY again."""
        if self._side_active and self.group_wgrad and self.fuse_bias_grad:
            for i in range(len(self._deferred) - 1, -1, -1):
                it = self._deferred[i]
                if (isinstance(it, tuple) and it[0] == 'dw' and len(it) == 15 and it[2].data_ptr() == x.data_ptr() and it[4] == M
                        and it[8] == 1 and it[9] == 0 and 0 <= C - it[7] < 4 and it[11] == (ldx or x.stride(-2))):
                    self._deferred[i] = it + (out,)
                    return
        self._side(('cs', x, ldx or x.stride(-2), out, M, C))

    def dense_fwd(self, x, scope, y, M, cin, cout, act=0):
        self.gemm(x, self.P(scope + '/kernel'), self.P(scope + '/bias'), y, M, cin, cout, act=act)

    def dense_bwd(self, x, dy, scope, M, cin, cout, dx=None, acc=0):
        """dy = gradient wrt the layer's pre-activation."""
        self.gemm_dw(x, dy, self.G(scope + '/kernel'), M, cin, cout)
        self.colsum(dy, self.G(scope + '/bias'), M, cout)
        if dx is not None:
            self.gemm_dx(dy, self.P(scope + '/kernel'), dx, M, cin, cout, acc=acc)

    # ---- conv + BN ------------------------------------------------------------------------------------------------
    def bn_fwd(self, scope, x, M, C, training, dstat=None):
        """dstat: the per-column sums were already accumulated (taco_bn_stats_rows, tensor produced in pieces): finalise only."""
        sc, sh = self.buf(scope + '/bn_scale', C), self.buf(scope + '/bn_shift', C)
        if training and dstat is not None:
            lib.taco_bn_finalize(dstat, self.P(scope + '/gamma'), self.P(scope + '/beta'),
                                 self.L.bnview(self.bnbatch, scope + '/moving_mean'),
                                 self.L.bnview(self.bnbatch, scope + '/moving_variance'),
                                 self.buf(scope + '/bn_rstd', C), sc, sh, M, C, BN_EPS, self.st)
        elif training:
            lib.taco_bn_stats_fwd(x, x.stride(-2), self.P(scope + '/gamma'), self.P(scope + '/beta'), self.dslot(24 * C),
                                  self.L.bnview(self.bnbatch, scope + '/moving_mean'),
                                  self.L.bnview(self.bnbatch, scope + '/moving_variance'),
                                  self.buf(scope + '/bn_rstd', C), sc, sh, M, C, BN_EPS, self.st)
        else:
            lib.taco_bn_infer_params(self.L.bnview(self.bn, scope + '/moving_mean'),
                                     self.L.bnview(self.bn, scope + '/moving_variance'),
                                     self.P(scope + '/gamma'), self.P(scope + '/beta'), sc, sh, C, BN_EPS, self.st)
        return sc, sh

    def bn_bwd(self, scope, x, dy, dx, M, C, T, pool, relu):
        lib.taco_bn_bwd(x, x.stride(-2), dy, dy.stride(-2), self.L.bnview(self.bnbatch, scope + '/moving_mean'),
                        self.buf(scope + '/bn_rstd', C), self.buf(scope + '/bn_scale', C), self.buf(scope + '/bn_shift', C),
                        self.P(scope + '/gamma'), self.dslot(24 * C), self.G(scope + '/gamma'), self.G(scope + '/beta'),
                        self.G(scope + '/bias'), dx, dx.stride(-2), M, C, T, pool, relu, self.st)

    # ---- CBHG (models/modules.py:35-74) ----------------------------------------------------------------------------
    # ---- frame-band pipelines behind the post-net biGRU (round 3) -------------------------------------------------------------
    # The post-net biGRU is 640 dependent steps on 64 of 256 CUs, forward and again in BPTT (~0.5 ms each at C2), and what follows
    # it is row-independent: linear layer -> L1 loss -> linear input gradient after the forward pass, input projection gradient ->
    # highway stack BPTT after the backward pass.  Both directions sweep the whole sequence, so once the recurrence is past the
    # middle, step p completes the frames [T - p, p): the recurrence is cut into a few chunk launches (state carried through a
    # small buffer) and after each chunk the consumers of the frames it completed run on the auxiliary stream beside the next chunk.
    # Only the last band (the sequence ends) is left when the recurrence finishes.

    def _tail_chunks(self, T, training):
        fr = os.environ.get('TACO_TAIL_PLAN', '0')          # measured: no gain at C2, off by default
        if not training or T < int(os.environ.get('TACO_TAIL_MIN_T', '256')) or fr in ('', '0'):
            return [(0, T)]
        cuts = [0]
        for f in fr.split(':'):
            c = min(T, int(round(T * float(f) / 8)) * 8)
            if c > cuts[-1] and c < T:
                cuts.append(c)
        cuts.append(T)
        return list(zip(cuts[:-1], cuts[1:]))

    @staticmethod
    def _new_bands(T, q, p):
        """frame ranges whose two directions are both done after step p, and were not after step q (q < p)"""
        lo, hi = T - p, p
        if hi <= lo:
            return []
        qlo, qhi = T - q, q
        if qhi <= qlo:
            return [(lo, hi)]
        return [bd for bd in ((lo, qlo), (qhi, hi)) if bd[1] > bd[0]]

    def _chunked_bigru(self, launch, T, chunks, tail):
        """launch(s0, s1, pad) runs one chunk on the current stream; tail(bands) the consumers of completed frames on the auxiliary
        stream.  Returns after the current stream has been made to wait for the auxiliary stream."""
        cur = torch.cuda.current_stream()
        aux = self.stream_d
        pad = int(os.environ.get('TACO_TAIL_PAD', '150000')) if len(chunks) > 1 else 0
        for (s0, s1) in chunks:
            launch(s0, s1, pad)
            bands = self._new_bands(T, s0, s1)
            if tail is not None and bands:
                ev = torch.cuda.Event(); ev.record(cur)
                aux.wait_event(ev)
                with torch.cuda.stream(aux):
                    tail(bands)
        if tail is not None:
            cur.wait_stream(aux)

    def cbhg_fwd(self, sc, x, N, T, cin, K, proj, lengths, training, bank_dstat=None, tail=None):
        """bank_dstat: the conv bank output (buffer sc/bank) and its batch-norm sums were produced piecewise by the caller."""
        M, C, st = N * T, K * 128, self.st
        B = self.buf(sc + '/bank', M, C)
        # training: the batch-norm sums of every conv output are taken in that conv's GEMM epilogue (conv_bn), not by a pass over it
        if bank_dstat is None and training:
            bank_dstat = self.conv_bn(x, self.P(sc + '/conv_bank/kernel'), self.P(sc + '/conv_bank/bias'), B, M, cin, C, T, kw=K, bank=K,
                                      ldw=128, act=ACT_RELU)
        elif bank_dstat is None:
            self.gemm(x, self.P(sc + '/conv_bank/kernel'), self.P(sc + '/conv_bank/bias'), B, M, cin, C, T=T, kw=K, bank=K,
                      ldw=128, act=ACT_RELU)
        s, h = self.bn_fwd(sc + '/conv_bank', B, M, C, training, dstat=bank_dstat)
        PL = self.buf(sc + '/pool', M, C)
        lib.taco_bn_apply_fwd(B, C, s, h, None, 0, PL, C, M, C, T, 1, st)
        C1 = self.buf(sc + '/c1', M, proj[0])
        d1 = None
        if training:
            d1 = self.conv_bn(PL, self.P(sc + '/proj_1/kernel'), self.P(sc + '/proj_1/bias'), C1, M, C, proj[0], T, kw=3, act=ACT_RELU)
        else:
            self.gemm(PL, self.P(sc + '/proj_1/kernel'), self.P(sc + '/proj_1/bias'), C1, M, C, proj[0], T=T, kw=3, act=ACT_RELU)
        s, h = self.bn_fwd(sc + '/proj_1', C1, M, proj[0], training, dstat=d1)
        Y1 = self.buf(sc + '/y1', M, proj[0])
        lib.taco_bn_apply_fwd(C1, proj[0], s, h, None, 0, Y1, proj[0], M, proj[0], T, 0, st)
        C2 = self.buf(sc + '/c2', M, proj[1])
        d2 = None
        if training:
            d2 = self.conv_bn(Y1, self.P(sc + '/proj_2/kernel'), self.P(sc + '/proj_2/bias'), C2, M, proj[0], proj[1], T, kw=3)
        else:
            self.gemm(Y1, self.P(sc + '/proj_2/kernel'), self.P(sc + '/proj_2/bias'), C2, M, proj[0], proj[1], T=T, kw=3)
        s, h = self.bn_fwd(sc + '/proj_2', C2, M, proj[1], training, dstat=d2)
        HW0 = self.buf(sc + '/hw0', M, proj[1])
        lib.taco_bn_apply_fwd(C2, proj[1], s, h, x, cin, HW0, proj[1], M, proj[1], T, 0, st)   # + residual
        hw = HW0
        if proj[1] != 128:
            hw = self.buf(sc + '/hwd', M, 128)
            self.dense_fwd(HW0, sc + '/highway_dense', hw, M, proj[1], 128)
        Zs = [self.buf('%s/hwZ%d' % (sc, i), M, 256) for i in range(1, 5)]
        ys = [self.buf('%s/hw%d' % (sc, i), M, 128) for i in range(1, 5)]
        if self.fused_highway and M >= self.FUSED_HIGHWAY_MIN_ROWS:
            # the four layers in one launch: the tile's activations stay in LDS, [W_H | W_T] streams through an LDS ring
            pa = lambda ts: (ctypes.c_void_p * 4)(*[t.data_ptr() for t in ts])
            self._timed('highway x4 fwd (highway4_fwd_k)', 4 * 2.0 * M * 128 * 256,
                        lambda: lib.taco_highway4_fwd(hw, pa([self.P('%s/highway_%d/kernel' % (sc, i)) for i in range(1, 5)]),
                                                      pa([self.P('%s/highway_%d/bias' % (sc, i)) for i in range(1, 5)]), pa(Zs), pa(ys), M, st))
            hw = ys[3]
        else:
            for i in range(1, 5):
                self.gemm(hw, self.P('%s/highway_%d/kernel' % (sc, i)), self.P('%s/highway_%d/bias' % (sc, i)), Zs[i - 1], M, 128, 256)
                lib.taco_highway_gate_fwd(Zs[i - 1], hw, ys[i - 1], M, st)
                hw = ys[i - 1]
        XP = self.buf(sc + '/xp', M, 768)
        self.gemm(hw, self.P(sc + '/bigru/wx'), self.P(sc + '/bigru/bias'), XP, M, 128, 768)
        OUT = self.buf(sc + '/out', M, 256)
        RUC = self.buf(sc + '/ruc', 2, N, T, 384)
        chunks = self._tail_chunks(T, training) if tail is not None else [(0, T)]
        state = self.buf(sc + '/gru_state', 2, N, 128)
        self._chunked_bigru(
            lambda s0, s1, pad: self._timed(
                'biGRU(128) fwd (gru128_seq_fwd_k)', 2.0 * 2 * N * (s1 - s0) * 128 * 384,
                lambda: lib.taco_gru128_seq_fwd(XP, 768, self.P(sc + '/bigru/fw_whg'), self.P(sc + '/bigru/fw_whc'),
                                                self.P(sc + '/bigru/bw_whg'), self.P(sc + '/bigru/bw_whc'), lengths, OUT, 256, RUC, N, T, 2,
                                                s0, s1, state, pad, self.st)),
            T, chunks, tail)
        return OUT

    def cbhg_bwd(self, sc, x, dOUT, N, T, cin, K, proj, lengths, dx, eager=False, after_proj2=None, bank_dx_later=False, tail=False):
        """dOUT [M,256] gradient wrt the CBHG output; writes the gradient wrt the CBHG input x into dx [M,cin].
        eager: release the deferred weight-gradient launches to the side stream after every block (encoder: nothing
        latency-bound follows that they could disturb, and held back they would run as a serial tail after the main stream)."""
        # release points: 0 after the highway stack, 1 after proj_2, 2 after proj_1, 3 after the bank (TACO_ENC_FLUSH: which of them the
        # eager mode uses; default only the last: fewer, larger grouped launches, 7.55 -> 7.52 ms against all four)
        sites = self.enc_flush_sites if eager else ()
        flush = lambda k: self.flush_side() if k in sites else None
        M, C, st = N * T, K * 128, self.st
        b = self._bufs
        OUT, RUC = b[sc + '/out'], b[sc + '/ruc']
        dXP = self.buf(sc + '/dxp', M, 768)
        HP, RH = self.buf(sc + '/hp', 2, M, 128), self.buf(sc + '/rh', 2, M, 128)
        ready = self.inputs_ready() if self._deferred else None
        dhw = self.buf(sc + '/dhw_a', M, 128)
        other = self.buf(sc + '/dhw_b', M, 128)
        hw_ins = [(b[sc + '/hwd'] if proj[1] != 128 else b[sc + '/hw0'])] + [b['%s/hw%d' % (sc, i)] for i in range(1, 4)]
        dZs = [self.buf('%s/dZ%d' % (sc, i), M, 256) for i in range(1, 5)]      # one per layer: read later by the side-stream dW GEMM
        pa = lambda ts: (ctypes.c_void_p * 4)(*[t.data_ptr() for t in ts])
        chunks = self._tail_chunks(T, True) if (tail and self.fused_highway and M >= self.FUSED_HIGHWAY_MIN_ROWS) else [(0, T)]
        banded = len(chunks) > 1
        dHW0_b = self.buf(sc + '/dhw0', M, proj[1]) if (banded and proj[1] != 128) else None

        def btail(bands):
            # frames whose BPTT is complete in BOTH directions: gradient wrt the highway output, the highway stack's BPTT and (post-net)
            # the 80 -> 128 dense layer's input gradient, all row-independent
            for (f0, f1) in bands:
                self._timed('dX GEMM (conv_gemm_nt2)', 2.0 * N * (f1 - f0) * 128 * 768,
                            lambda: lib.taco_dense_rows_bwd_data(dXP, self.P(sc + '/bigru/wx'), dhw, N, T, f0, f1, 128, 768, 768, 768, 128, 0, self.st))
                self._timed('highway x4 bwd (highway4_bwd_k)', 4 * 2.0 * N * (f1 - f0) * 128 * 256,
                            lambda: lib.taco_highway4_bwd_rows(dhw, pa([b['%s/hwZ%d' % (sc, i)] for i in range(1, 5)]), pa(hw_ins),
                                                               pa([self.P('%s/highway_%d/kernel' % (sc, i)) for i in range(1, 5)]), pa(dZs), other,
                                                               N, T, f0, f1, self.st))
                if dHW0_b is not None:
                    lib.taco_dense_rows_bwd_data(other, self.P(sc + '/highway_dense/kernel'), dHW0_b, N, T, f0, f1, proj[1], 128, 128, 128,
                                                 proj[1], 0, self.st)
        state = self.buf(sc + '/gru_bstate', 2, N, 128)
        first = [True]

        def launch(s0, s1, pad):
            if sc == 'encoder_cbhg' and not pad:
                # the encoder's BPTT runs beside the decoder's weight-gradient GEMMs: with its CUs to itself (150 KB of unused LDS, like the
                # decoder GRUs) it takes ~100 us instead of ~250 on the main stream's chain; C2 6.62 -> 6.56 ms (100 KB: a GEMM workgroup
                # still fits beside it, no gain).  Not with more than one rank: the exchange's kernels need CUs too (_gru256_pad).
                pad = int(os.environ.get('TACO_ENC_GRU_PAD', '150000' if self.world == 1 else '0'))
            self._timed('biGRU(128) bwd (gru128_seq_bwd_k)', 2.0 * 2 * N * (s1 - s0) * 128 * 384,
                        lambda: lib.taco_gru128_seq_bwd(dOUT, 256, self.P(sc + '/bigru/fw_whg'), self.P(sc + '/bigru/fw_whc'),
                                                        self.P(sc + '/bigru/bw_whg'), self.P(sc + '/bigru/bw_whc'), lengths, OUT, 256, RUC,
                                                        dXP, 768, HP, RH, N, T, 2, s0, s1, state, pad, self.st))
            if first[0]:
                self.flush_side(ready)             # pending weight gradients start beside the first chunk
                first[0] = False
        self._chunked_bigru(launch, T, chunks, btail if banded else None)
        hw4 = b[sc + '/hw4']
        self.gemm_dw(hw4, dXP, self.G(sc + '/bigru/wx'), M, 128, 768)
        self.colsum(dXP, self.G(sc + '/bigru/bias'), M, 768)
        for di, d in enumerate(('fw', 'bw')):
            self.gemm_dw(HP[di], dXP[:, di * 384:], self.G('%s/bigru/%s_whg' % (sc, d)), M, 128, 256, ldx=128, lddy=768, ldw=256)
            self.gemm_dw(RH[di], dXP[:, di * 384 + 256:], self.G('%s/bigru/%s_whc' % (sc, d)), M, 128, 128, ldx=128, lddy=768, ldw=128)
        if not banded:
            self.gemm_dx(dXP, self.P(sc + '/bigru/wx'), dhw, M, 128, 768)
        if banded:
            for i in range(4, 0, -1):
                self.gemm_dw(hw_ins[i - 1], dZs[i - 1], self.G('%s/highway_%d/kernel' % (sc, i)), M, 128, 256)
                self.colsum(dZs[i - 1], self.G('%s/highway_%d/bias' % (sc, i)), M, 256)
            dhw, other = other, dhw
            flush(0)
        elif self.fused_highway and M >= self.FUSED_HIGHWAY_MIN_ROWS:
            pa = lambda ts: (ctypes.c_void_p * 4)(*[t.data_ptr() for t in ts])
            self._timed('highway x4 bwd (highway4_bwd_k)', 4 * 2.0 * M * 128 * 256,
                        lambda: lib.taco_highway4_bwd(dhw, pa([b['%s/hwZ%d' % (sc, i)] for i in range(1, 5)]), pa(hw_ins),
                                                      pa([self.P('%s/highway_%d/kernel' % (sc, i)) for i in range(1, 5)]), pa(dZs), other, M, st))
            for i in range(4, 0, -1):
                self.gemm_dw(hw_ins[i - 1], dZs[i - 1], self.G('%s/highway_%d/kernel' % (sc, i)), M, 128, 256)
                self.colsum(dZs[i - 1], self.G('%s/highway_%d/bias' % (sc, i)), M, 256)
            dhw, other = other, dhw
            flush(0)
        else:
            for i in range(4, 0, -1):
                dZ, hw_in = dZs[i - 1], hw_ins[i - 1]
                lib.taco_highway_gate_bwd(b['%s/hwZ%d' % (sc, i)], hw_in, dhw, dZ, other, M, st)
                self.gemm_dw(hw_in, dZ, self.G('%s/highway_%d/kernel' % (sc, i)), M, 128, 256)
                self.colsum(dZ, self.G('%s/highway_%d/bias' % (sc, i)), M, 256)
                self.gemm_dx(dZ, self.P('%s/highway_%d/kernel' % (sc, i)), other, M, 128, 256, acc=1)
                dhw, other = other, dhw
                flush(0)
        if proj[1] != 128:
            dHW0 = self.buf(sc + '/dhw0', M, proj[1])
            # (banded: the input gradient was computed band by band above; the weight / bias gradients are deferred as always)
            self.dense_bwd(b[sc + '/hw0'], dhw, sc + '/highway_dense', M, proj[1], 128, dx=None if banded else dHW0)
        else:
            dHW0 = dhw
        # proj_2 (no activation) -> proj_1 (relu) -> pooled bank (relu)
        dC2 = self.buf(sc + '/dc2', M, proj[1])
        self.bn_bwd(sc + '/proj_2', b[sc + '/c2'], dHW0, dC2, M, proj[1], T, 0, 0)
        self.gemm_dw(b[sc + '/y1'], dC2, self.G(sc + '/proj_2/kernel'), M, proj[0], proj[1], T=T, kw=3)
        dY1 = self.buf(sc + '/dy1', M, proj[0])
        self.gemm_dx(dC2, self.P(sc + '/proj_2/kernel'), dY1, M, proj[0], proj[1], T=T, kw=3)
        if after_proj2 is not None:
            after_proj2()              # gradients of the biGRU, the highways and proj_2 are all created: bucket boundary
        flush(1)
        dC1 = self.buf(sc + '/dc1', M, proj[0])
        self.bn_bwd(sc + '/proj_1', b[sc + '/c1'], dY1, dC1, M, proj[0], T, 0, 1)
        self.gemm_dw(b[sc + '/pool'], dC1, self.G(sc + '/proj_1/kernel'), M, C, proj[0], T=T, kw=3)
        dPL = self.buf(sc + '/dpool', M, C)
        self.gemm_dx(dC1, self.P(sc + '/proj_1/kernel'), dPL, M, C, proj[0], T=T, kw=3)
        flush(2)
        dB = self.buf(sc + '/dbank', M, C)
        self.bn_bwd(sc + '/conv_bank', b[sc + '/bank'], dPL, dB, M, C, T, 1, 1)
        self.gemm_dw(x, dB, self.G(sc + '/conv_bank/kernel'), M, cin, C, T=T, kw=K, bank=K, ldw=128)
        if bank_dx_later:                  # the caller computes the bank's input gradient piecewise (and adds the residual dHW0)
            flush(3)
            return dB, dHW0
        self.gemm_dx(dB, self.P(sc + '/conv_bank/kernel'), dx, M, cin, C, T=T, kw=K, bank=K, ldw=128)
        flush(3)
        lib.taco_add(dx, dHW0, dx, M * cin, 0, st)        # residual connection (modules.py:56)
        return dx

    # ---- forward (models/tacotron.py:35-104) ------------------------------------------------------------------------
    def forward(self, inputs, input_lengths, mel_targets, identities=None, training=True, linear_targets=None):
        """linear_targets (training): the linear layer, its L1 loss and the loss gradient back to the post-net output are then
        computed band by band behind the post-net biGRU (see _tail_chunks); loss() only adds the mel loss."""
        L, st = self.L, self.st
        N, Ti = inputs.shape
        To = mel_targets.shape[1]
        r, nm = self.r, self.nm
        assert To % r == 0, 'T_out must be a multiple of outputs_per_step (datafeeder_npy.py:179-181)'
        S = To // r
        self.dims = (N, Ti, To, S)
        self._dpos = 0
        lib.taco_zero(self._zbuf, self._zbuf.numel(), st)      # gradients + reduction scratch: one memset node
        self.inputs, self.input_lengths, self.mel_targets, self.identities = inputs, input_lengths, mel_targets, identities
        E = L.Et + L.Es
        Me, Mp, Ms = N * Ti, N * To, N * S
        self._mark('start')
        X0 = self.buf('emb', Me, E)
        lib.taco_embed_gather_fwd(inputs, identities if L.Es else None, self.P('embedding'),
                                  self.P('embedding_id') if L.Es else None, X0, N, Ti, L.Et, L.Es, L.vocab, max(L.id_num, 1), st)
        A1, A2 = self.buf('enc_p1', Me, 256), self.buf('enc_p2', Me, 128)
        self.dense_fwd(X0, 'prenet/dense_1', A1, Me, E, 256, ACT_RELU)
        self.dense_fwd(A1, 'prenet/dense_2', A2, Me, 256, 128, ACT_RELU)
        ENC = self.cbhg_fwd('encoder_cbhg', A2, N, Ti, 128, 16, (128, 128), input_lengths, training)
        self._mark('encoder fwd')
        # ---- decoder
        KEYS = self.buf('keys', Me, 256)
        self.gemm(ENC, self.P('attention/memory_layer/kernel'), None, KEYS, Me, 256, 256)
        FR = self.buf('frames', Ms, nm)
        lib.taco_gather_frames(mel_targets, FR, N, S, r, nm, st)
        W1 = self.P('decoder_prenet/dense_1/kernel')
        F1 = self.buf('F1', Ms, 256)
        self.gemm(FR, W1, self.P('decoder_prenet/dense_1/bias'), F1, Ms, nm, 256)
        z = self.buf('zeros', max(N, 32) * 256)
        self._attn_ptrs = self._make_attn_ptrs(N, S, Ti)
        HC = self._bufs['HC']
        Y = self.buf('Y', Ms, 256)
        gb = {}
        for g in (1, 2):
            gb[g] = dict(XP=self.buf('xp%d' % g, Ms, 768), D=self.buf('D%d' % g, Ms, 256),
                         t=[self.buf('g%d_%s' % (g, k), Ms, 256) for k in ('r', 'u', 'c', 'rh', 'h')],
                         x=self.buf('xchg_g%d' % g, ((N + self.GRU256_ROWS - 1) // self.GRU256_ROWS) * self.GRU256_XCHG, dtype=torch.int64))
        # Chunk-pipelined decoder: attention recurrence on the current stream, GRU1 / GRU2 on two more streams; chunk c
        # of GRU1 (its hoisted projections first) starts as soon as the attention kernel has finished chunk c.
        chunks = self._chunks(N, S, Ti)
        cur = torch.cuda.current_stream()
        sb, sc_ = (self.stream_b, self.stream_c) if len(chunks) > 1 else (cur, cur)
        Wp, bp = self.P('concat_projection/kernel'), self.P('concat_projection/bias')
        MEL = self.buf('mel_out', N, To, nm)             # == decoder outputs [N,S,nm*r] (tacotron.py:97)
        # Post-net conv bank behind the decoder pipeline: as soon as GRU2 has finished a chunk, its output projection, the bank conv
        # of the frames that chunk completes (a conv of width k reaches k // 2 frames ahead) and their batch-norm sums run on the
        # GRU2 stream, on the CUs the attention recurrence leaves idle; after the last chunk only its share (~14 %) is left.
        pipe_post = self.post_pipe and training and len(chunks) > 1
        if pipe_post:
            Kp, Cp = 8, 8 * 128
            Bpost, dst_post, done = self.buf('post_cbhg/bank', Mp, Cp), self.dslot(24 * Cp), 0
            Wo, bo = self.P('output_projection/kernel'), self.P('output_projection/bias')
        for ci, (s0, s1) in enumerate(chunks):
            for nb, tab in self._attn_ptrs:
                self._timed('attention recurrence fwd (attn_cluster_fwd_k)', self._attn_flops(nb, Ti, s1 - s0),
                            lambda: lib.taco_attn_rnn_fwd(tab, self._dims(nb, S, Ti, s0, s1), self.st))
            ev = torch.cuda.Event(); ev.record(cur)
            sb.wait_event(ev)
            with torch.cuda.stream(sb):
                self.dense_rows(HC, Wp, bp, Y, N, S, s0, s1, 512, 256, 512, 256)
                self.dense_rows(Y, self.P('decoder_gru_1/wx'), self.P('decoder_gru_1/bias'), gb[1]['XP'], N, S, s0, s1, 256, 768, 256, 768)
                t = gb[1]['t']
                self.gru256_fwd(gb[1]['XP'], self.P('decoder_gru_1/whg'), self.P('decoder_gru_1/whc'), Y, t, gb[1]['D'], gb[1]['x'],
                                N, S, s0, s1)
                ev2 = torch.cuda.Event(); ev2.record(sb)
            sc_.wait_event(ev2)
            with torch.cuda.stream(sc_):
                self.dense_rows(gb[1]['D'], self.P('decoder_gru_2/wx'), self.P('decoder_gru_2/bias'), gb[2]['XP'], N, S, s0, s1, 256, 768, 256, 768)
                t = gb[2]['t']
                self.gru256_fwd(gb[2]['XP'], self.P('decoder_gru_2/whg'), self.P('decoder_gru_2/whc'), gb[1]['D'], t, gb[2]['D'],
                                gb[2]['x'], N, S, s0, s1)
                if pipe_post:
                    ev3 = torch.cuda.Event(); ev3.record(sc_)
            if pipe_post:
                # on a stream of its own: queued on the GRU2 stream the pieces would sit in front of GRU2's next chunk
                sd = self.stream_d
                sd.wait_event(ev3)
                with torch.cuda.stream(sd):
                    self.dense_rows(gb[2]['D'], Wo, bo, MEL.view(Ms, nm * r), N, S, s0, s1, 256, nm * r, 256, nm * r)
                    f1 = To if ci == len(chunks) - 1 else r * s1 - Kp // 2
                    if f1 > done:
                        f0 = done
                        self._timed('fwd GEMM (conv_gemm_nn2)', self._gemm_flops(N * (f1 - f0), nm, Cp, Kp, Kp),
                                    lambda: lib.taco_conv_rows_fwd(MEL.view(Mp, nm), self.P('post_cbhg/conv_bank/kernel'),
                                                                   self.P('post_cbhg/conv_bank/bias'), Bpost, N, To, f0, f1, nm, Cp, Kp, Kp,
                                                                   nm, 128, Cp, ACT_RELU, dst_post, self.st))
                        done = f1
        if len(chunks) > 1:
            cur.wait_stream(sb); cur.wait_stream(sc_)
        if pipe_post:
            cur.wait_stream(self.stream_d)
        if not pipe_post:
            self.dense_fwd(gb[2]['D'], 'output_projection', MEL.view(Ms, nm * r), Ms, 256, nm * r)
        self._mark('decoder fwd')
        LIN = self.buf('lin_out', N, To, self.nf)
        self._tail_done = False
        tail = None
        if training and linear_targets is not None and len(self._tail_chunks(To, training)) > 1:
            # linear layer (tacotron.py:101) + linear L1 loss (:132-135) + gradient wrt the post-net output, per band of frames
            self.loss_sums = self.dslot(32)
            dLIN, dPOST = self.buf('dlin', Mp, L.ld_lin), self.buf('dpost', Mp, 256)
            Wl, bl = self.P('linear/kernel'), self.P('linear/bias')
            w_all, w_pri = 0.5 / (Mp * self.nf), 0.5 / (Mp * self.npri)

            def tail(bands):
                POSTb = self._bufs['post_cbhg/out']
                for (f0, f1) in bands:
                    rows = N * (f1 - f0)
                    self._timed('fwd GEMM (conv_gemm_nn2)', 2.0 * rows * 256 * self.nf,
                                lambda: lib.taco_dense_rows_fwd(POSTb, Wl, bl, LIN, N, To, f0, f1, 256, self.nf, 256, L.ld_lin, self.nf, 0, 0, self.st))
                    lib.taco_l1_loss_rows(LIN, self.nf, linear_targets, self.nf, dLIN, L.ld_lin, self.loss_sums[16:], N, To, f0, f1,
                                          self.nf, self.npri, w_all, w_pri, self.st)
                    self._timed('dX GEMM (conv_gemm_nt2)', 2.0 * rows * 256 * L.ld_lin,
                                lambda: lib.taco_dense_rows_bwd_data(dLIN, Wl, dPOST, N, To, f0, f1, 256, L.ld_lin, L.ld_lin, L.ld_lin, 256, 0, self.st))
            self.buf('post_cbhg/out', Mp, 256)               # exists before the closure runs
            self._tail_done = True
        POST = self.cbhg_fwd('post_cbhg', MEL.view(Mp, nm), N, To, nm, 8, (256, nm), None, training,
                             bank_dstat=dst_post if pipe_post else None, tail=tail)
        if not self._tail_done:
            self.gemm(POST, self.P('linear/kernel'), self.P('linear/bias'), LIN, Mp, 256, self.nf, ldw=L.ld_lin, ldy=self.nf)
        self._mark('post-net fwd')
        self.mel_outputs, self.linear_outputs = MEL, LIN
        self.alignments = self._bufs['ALIGN'].view(N, S, Ti).transpose(1, 2)      # [N, Ti, S] (tacotron.py:104)
        self.encoder_outputs = ENC.view(N, Ti, 256)
        return MEL, LIN, self.alignments

    RCCL_RESERVE_CUS = 32          # CUs the persistent kernels leave to the collective kernels under data parallelism

    def _persistent_wgs(self, N):
        """(attention, GRU(256)) workgroups of one chunk launch: 8 per two rows of an attention row block, 4 per two rows of a GRU
        row block."""
        return 8 * ((min(N, self.ATTN_ROWS) + 1) // 2), 4 * ((min(N, self.GRU256_ROWS) + 1) // 2)

    def _gru256_pad(self, N):
        """Isolation pad of the GRU(256) cluster kernels (csrc/gru256.hip): 150 KB of unused LDS give a workgroup its CU to itself,
        which pays while every persistent workgroup of the pipeline has a CU of its own AND nothing else needs room there: not for
        N > 32, and not under data parallelism, where the RCCL kernels must find LDS and registers beside the GRU workgroups (the
        attention workgroups fill their CUs' register files, so the 4 N attention CUs are closed to everything else anyway).
        TACO_GRU256_PAD=<bytes> overrides."""
        forced = os.environ.get('TACO_GRU256_PAD')
        if forced is not None:
            return int(forced)
        attn, gru = self._persistent_wgs(N)
        return 150000 if (self.world == 1 and attn + 2 * gru <= 256) else 0

    def _check_residency(self, N, Ti):
        """Co-residency of the persistent workgroups the chunk pipeline keeps in flight: one attention launch, one GRU1 and one GRU2
        launch on three streams (launches of one kind serialise on their stream).  All of them spin on cluster peers, so a launch
        makes progress only once ALL its workgroups are placed.  One CU holds ONE of these workgroups: 512 threads = 8 waves, 2 per
        SIMD, at 249-256 VGPRs (attention: the register file is full, nothing else fits on that CU) or 154-170 VGPRs (GRU(256): a
        second 8-wave workgroup would need 4 waves per SIMD; waves of up to ~160 VGPRs of OTHER kernels still fit beside it unless
        the isolation pad closes the LDS).  Rule: attention + 2 x GRU workgroups <= 256 CUs minus, under data parallelism, the CUs
        reserved for the collective kernels (RCCL_RESERVE_CUS; they can also share the unpadded GRU CUs).  C2: 128 + 64 + 64 = 256
        at world 1 (every CU owned); world > 1 runs unpadded.  Batches beyond that rule are NOT pipelined (_chunks returns one
        chunk: a single stream, one persistent launch at a time, <= 256 workgroups each by the row-block limits), because two
        partially placed launches could hold the CUs each other needs until the bounded spins give up (err word).  Side-stream GEMM
        workgroups never spin, so they can delay a cluster workgroup's dispatch but not deadlock it."""
        attn, gru = self._persistent_wgs(N)
        budget = 256 - (self.RCCL_RESERVE_CUS if (self.world > 1 and self._gru256_pad(N) > 0) else 0)
        if attn > 256 or gru > 256:
            raise RuntimeError('persistent cluster kernels would not be co-resident: %d / %d workgroups in one launch' % (attn, gru))
        return attn + 2 * gru <= budget

    def _chunks(self, N, S, Ti, k=None, plan_env='TACO_CHUNK_PLAN'):
        """Step ranges for the chunk-pipelined decoder (needs the cluster path); [(0, S)] = no pipelining."""
        k = k or self.pipe_chunks
        if k <= 1 or S < 2 * k or self.no_cluster or not lib.load().taco_attn_cluster_supported(min(N, self.ATTN_ROWS), Ti):
            return [(0, S)]
        if not self._check_residency(N, Ti) and os.environ.get('TACO_PIPE_OVERSUBSCRIBE', '0') != '1':
            return [(0, S)]          # the three launches of a pipeline stage would not all be resident: run them one at a time
        plan = os.environ.get(plan_env, '') or os.environ.get('TACO_CHUNK_PLAN', '')
        if not plan and plan_env == 'TACO_CHUNK_PLAN_BWD' and S >= 120 and k == 4:
            # BPTT processes the chunks last-in-time first: a short first chunk (GRU2 -> GRU1 -> attention lead-in) and growing ones
            # behind it.  Measured (scripts/dev_ab.py, round 3): C2 (S 128) 7.07 -> 7.01 ms, C5 (S 400) 12.04 -> 11.68 ms against the
            # forward plan; C4 (S 96) 5.32 -> 5.38 ms, so shorter sequences keep the forward plan.
            plan = '48:44:26:10'
        if plan:                                       # explicit relative chunk lengths, e.g. "40,36,28,16,8" (tuning aid)
            w = [float(x) for x in plan.split(':')]
            cuts = [0]
            for x in w[:-1]:
                cuts.append(min(S, max(cuts[-1] + 1, int(round(cuts[-1] + S * x / sum(w))))))
            cuts.append(S)
            return [(cuts[i], cuts[i + 1]) for i in range(len(w)) if cuts[i + 1] > cuts[i]]
        # The last chunk is half as long as the others: GRU1/GRU2 of the last chunk run after the attention recurrence has
        # finished (forward), and GRU2/GRU1 of the last chunk run before the attention BPTT can start (backward).
        last = max(1, int(S * self.last_chunk_frac / k))
        step = (S - last + k - 2) // (k - 1)
        bounds = [min(S - last, i * step) for i in range(k)] + [S]
        return [(bounds[i], bounds[i + 1]) for i in range(k) if bounds[i + 1] > bounds[i]]

    @staticmethod
    def _attn_flops(N, Ti, steps):
        """forward FLOPs of the attention recurrence per launch (prenet context part + dense_2, GRU(256) on a
        128-wide input, query projection, 5*Ti*256 for the score / softmax / context tile)"""
        return steps * (2.0 * N * (256 * 256 + 256 * 128) + 2.0 * N * 384 * 768 + 2.0 * N * 256 * 256 + 5.0 * N * Ti * 256)

    def _dims(self, N, S, Ti, s0, s1):
        return (ctypes.c_int * 5)(N, S, Ti, s0, s1)

    # The GRU(256) cluster kernels hold <= 128 batch rows (4 workgroups per 2 rows on 256 CUs); rows are independent, so larger
    # batches run block by block on contiguous [n0:n1] row slices of the [N,S,*] tensors.
    GRU256_ROWS = 128
    # the fused highway kernels (32-row tiles) are used from the encoder's 4096 rows on (2 launches instead of 16, same time); smaller
    # problems are one tile's latency and stay on the per-layer GEMM + gate launches
    FUSED_HIGHWAY_MIN_ROWS = int(os.environ.get('TACO_FUSED_HIGHWAY_MIN_ROWS', '4096'))
    ATTN_ROWS = 64

    GRU256_XCHG = 64 * 2048        # granule slots of one GRU(256) row block (>= 64 clusters x (6 x 256 + 4))

    def gru256_fwd(self, xp, whg, whc, res, t, d, xchg, N, S, s0, s1):
        # every row block has its own granule region: the chunk launches of a pass share a buffer that is zero-filled once,
        # by the pass's first launch (epochs count the steps of the pass)
        for bi, n0 in enumerate(range(0, N, self.GRU256_ROWS)):
            n1 = min(N, n0 + self.GRU256_ROWS)
            v = lambda a, w: a.view(N, S * w)[n0:n1]
            xchg_b = xchg[bi * self.GRU256_XCHG:]
            self._timed('decoder GRU(256) fwd (gru256_cluster_fwd_k)', (s1 - s0) * 2.0 * (n1 - n0) * 256 * 768,
                        lambda: lib.taco_gru256_seq_fwd(v(xp, 768), whg, whc, v(res, 256), v(t[0], 256), v(t[1], 256), v(t[2], 256),
                                                        v(t[3], 256), v(t[4], 256), v(d, 256), xchg_b, self.err, n1 - n0, S, s0, s1,
                                                        self._gru256_pad(N), self.st))

    def gru256_bwd(self, dout, whg, whc, r, u, c, h, dxp, carry, xchg, N, S, s0, s1):
        for bi, n0 in enumerate(range(0, N, self.GRU256_ROWS)):
            n1 = min(N, n0 + self.GRU256_ROWS)
            v = lambda a, w: a.view(N, S * w)[n0:n1]
            xchg_b = xchg[bi * self.GRU256_XCHG:]
            self._timed('decoder GRU(256) bwd (gru256_cluster_bwd_k)', (s1 - s0) * 2.0 * (n1 - n0) * 256 * 768,
                        lambda: lib.taco_gru256_seq_bwd(v(dout, 256), whg, whc, v(r, 256), v(u, 256), v(c, 256), v(h, 256), v(dxp, 768),
                                                        carry.view(N, 256)[n0:n1], xchg_b, self.err, n1 - n0, S, s0, s1,
                                                        self._gru256_pad(N), self.st))

    def dense_rows(self, x, scope_w, bias, y, N, S, s0, s1, cin, cout, ldx, ldy):
        lib.taco_dense_rows_fwd(x, scope_w, bias, y, N, S, s0, s1, cin, cout, ldx, cout, ldy, 0, 0, self.st)

    def dense_rows_dx(self, dy, w, dx, N, S, s0, s1, cin, cout, lddy, lddx, acc):
        lib.taco_dense_rows_bwd_data(dy, w, dx, N, S, s0, s1, cin, cout, lddy, cout, lddx, acc, self.st)

    def _attn_slots(self, N, Ti):
        dll = lib.load()
        return max(dll.taco_attn_cluster_xchg_slots(N, Ti), dll.taco_attn_cluster_bwd_xchg_slots(N, Ti))

    _IP = ['W1', 'B1', 'W2', 'B2', 'WX', 'WHG', 'WHC', 'BG', 'WQ', 'V', 'KEYS', 'MEM', 'ZEROS', 'WP', 'BP', 'G1WX', 'G1B', 'G1WHG',
           'G1WHC', 'G2WX', 'G2B', 'G2WHG', 'G2WHC', 'WO', 'BO', 'HC', 'ALIGN', 'OUT', 'H1', 'H2', 'TMP', 'STOP']
    INFER_POLL = 100      # decoder steps enqueued between two reads of the device-side stop count

    def infer(self, inputs, input_lengths, identities=None, steps=None, max_iters=2000):
        """Free-running synthesis (reference models/tacotron.py:18-104 with linear_targets=None, models/helpers.py:7-38):
        batch norm in inference mode, the last predicted frame fed back, at most `steps` (default max_iters) decoder steps;
        decoding ends early once every row has produced an exactly-zero r-frame output (the reference's stop token, kept on
        the device and polled every INFER_POLL steps).  Returns (mel [N,S*r,80], linear, alignments [N,Ti,S]), S = steps run."""
        L, st = self.L, self.st
        N, Ti = inputs.shape
        S = int(steps if steps is not None else max_iters)
        r, nm = self.r, self.nm
        self._dpos = 0
        E = L.Et + L.Es
        Me = N * Ti
        X0 = self.buf('emb', Me, E)
        lib.taco_embed_gather_fwd(inputs, identities if L.Es else None, self.P('embedding'),
                                  self.P('embedding_id') if L.Es else None, X0, N, Ti, L.Et, L.Es, L.vocab, max(L.id_num, 1), st)
        A1, A2 = self.buf('enc_p1', Me, 256), self.buf('enc_p2', Me, 128)
        self.dense_fwd(X0, 'prenet/dense_1', A1, Me, E, 256, ACT_RELU)
        self.dense_fwd(A1, 'prenet/dense_2', A2, Me, 256, 128, ACT_RELU)
        ENC = self.cbhg_fwd('encoder_cbhg', A2, N, Ti, 128, 16, (128, 128), input_lengths, False)
        KEYS = self.buf('keys', Me, 256)
        self.gemm(ENC, self.P('attention/memory_layer/kernel'), None, KEYS, Me, 256, 256)
        z = self.buf('zeros', max(N, 32) * 256)
        P = self.P
        t = {'W1': P('decoder_prenet/dense_1/kernel'), 'B1': P('decoder_prenet/dense_1/bias'), 'W2': P('decoder_prenet/dense_2/kernel'),
             'B2': P('decoder_prenet/dense_2/bias'), 'WX': P('attention_gru/wx'), 'WHG': P('attention_gru/whg'), 'WHC': P('attention_gru/whc'),
             'BG': P('attention_gru/bias'), 'WQ': P('attention/query_layer/kernel'), 'V': P('attention/attention_v'), 'KEYS': KEYS, 'MEM': ENC,
             'ZEROS': z, 'WP': P('concat_projection/kernel'), 'BP': P('concat_projection/bias'),
             'G1WX': P('decoder_gru_1/wx'), 'G1B': P('decoder_gru_1/bias'), 'G1WHG': P('decoder_gru_1/whg'), 'G1WHC': P('decoder_gru_1/whc'),
             'G2WX': P('decoder_gru_2/wx'), 'G2B': P('decoder_gru_2/bias'), 'G2WHG': P('decoder_gru_2/whg'), 'G2WHC': P('decoder_gru_2/whc'),
             'WO': P('output_projection/kernel'), 'BO': P('output_projection/bias'),
             'HC': self.buf('i_HC', N * S, 512), 'ALIGN': self.buf('i_ALIGN', N * S, Ti), 'OUT': self.buf('i_mel', N, S * r, nm),
             'H1': self.buf('i_H1', 2, N, 256), 'H2': self.buf('i_H2', 2, N, 256), 'TMP': self.buf('i_TMP', 10, N, 256),
             'STOP': self.buf('i_STOP', 1 + N, dtype=torch.int32)}
        t['STOP'].zero_(); t['STOP'][0:1].fill_(S)
        arr = (ctypes.c_void_p * len(self._IP))(*[t[n].data_ptr() for n in self._IP])
        keep = S
        for s0 in range(0, S, self.INFER_POLL):
            s1 = min(S, s0 + self.INFER_POLL)
            lib.taco_decoder_infer(arr, (ctypes.c_int * 7)(N, S, Ti, r, nm, s0, s1), st)
            keep = int(t['STOP'][0].item())            # synchronises: the post-net needs the final length anyway
            if keep < S:
                break
        MEL = t['OUT']
        ALIGN = t['ALIGN'].view(N, S, Ti)
        if keep < S:                                    # every row hit the stop token: keep steps [0, keep)
            MEL = self.buf('i_mel_cut', N, keep * r, nm)
            MEL.copy_(t['OUT'].view(N, S, r * nm)[:, :keep].reshape(N, keep * r, nm))
            ALIGN = ALIGN[:, :keep]
            S = keep
        POST = self.cbhg_fwd('post_cbhg', MEL.view(N * S * r, nm), N, S * r, nm, 8, (256, nm), None, False)
        LIN = self.buf('i_lin', N, S * r, self.nf)
        self.gemm(POST, self.P('linear/kernel'), self.P('linear/bias'), LIN, N * S * r, 256, self.nf, ldw=L.ld_lin, ldy=self.nf)
        self.mel_outputs, self.linear_outputs = MEL, LIN
        self.alignments = ALIGN.transpose(1, 2)
        self.encoder_outputs = ENC.view(N, Ti, 256)
        return MEL, LIN, self.alignments

    def _make_attn_ptrs(self, N, S, Ti):
        b = self.buf
        W1 = self.P('decoder_prenet/dense_1/kernel')
        t = {
            'W1C': W1[self.nm:], 'F1': self._bufs['F1'], 'W2': self.P('decoder_prenet/dense_2/kernel'),
            'B2': self.P('decoder_prenet/dense_2/bias'), 'WX': self.P('attention_gru/wx'), 'WHG': self.P('attention_gru/whg'),
            'WHC': self.P('attention_gru/whc'), 'BG': self.P('attention_gru/bias'), 'WQ': self.P('attention/query_layer/kernel'),
            'V': self.P('attention/attention_v'), 'KEYS': self._bufs['keys'], 'MEM': self._bufs['encoder_cbhg/out'],
            'ZEROS': self._bufs['zeros'],
            'P1': b('P1', N * S, 256), 'P2': b('P2', N * S, 128), 'R': b('aR', N * S, 256), 'U': b('aU', N * S, 256),
            'C': b('aC', N * S, 256), 'RH': b('aRH', N * S, 256), 'HC': b('HC', N * S, 512), 'Q': b('Q', N * S, 256),
            'ALIGN': b('ALIGN', N * S, Ti),
            'DHC': b('dHC', N * S, 512), 'DXP': b('dXPa', N * S, 768), 'DP2': b('dP2', N * S, 128), 'DP1': b('dP1', N * S, 256),
            'DQ': b('dQ', N * S, 256), 'DKEYS': b('dKEYS', N * Ti, 256), 'DMEM': b('dMEM', N * Ti, 256),
            'DVPART': b('dVPART', N * Ti, 256), 'DA': b('dA', N * Ti), 'DHT': b('dHT', N, 256),
            'DHPART': b('dHPART', N, 256), 'DHCARRY': b('dHCARRY', N, 256), 'DCTX': b('dCTX', N, 256),
            'DCTXCARRY': b('dCTXCARRY', N, 256),
            'XCHG': None if self.no_cluster else b('xchg_attn', max(self._attn_slots(min(N, self.ATTN_ROWS), Ti), 8) *
                                                   ((N + self.ATTN_ROWS - 1) // self.ATTN_ROWS), dtype=torch.int64), 'ERR': self.err,
            'DE': b('dE', N * S, Ti), 'DCTXS': b('dCTXS', N * S, 256),
            'DAEXT': b('dALIGN_reg', N * S, Ti) if self.has_regularity else None,
        }
        # floats per batch row of every per-row tensor (weights / scratch shared by all rows: 0)
        row = dict(F1=S * 256, KEYS=Ti * 256, MEM=Ti * 256, P1=S * 256, P2=S * 128, R=S * 256, U=S * 256, C=S * 256, RH=S * 256,
                   HC=S * 512, Q=S * 256, ALIGN=S * Ti, DHC=S * 512, DXP=S * 768, DP2=S * 128, DP1=S * 256, DQ=S * 256,
                   DKEYS=Ti * 256, DMEM=Ti * 256, DVPART=Ti * 256, DA=Ti, DHT=256, DHPART=256, DHCARRY=256, DCTX=256,
                   DCTXCARRY=256, DE=S * Ti, DCTXS=S * 256, DAEXT=S * Ti)
        # The attention cluster kernels hold <= 64 batch rows per launch (8 workgroups per 2 rows on 256 CUs); rows are
        # independent, so a larger batch runs block by block on row-offset pointer tables.  (The per-step fallback kernels
        # take any N: one table.)
        blk = self.ATTN_ROWS if (not self.no_cluster and lib.load().taco_attn_cluster_supported(min(N, self.ATTN_ROWS), Ti)) else N
        tables = []
        xslots = max(self._attn_slots(min(N, self.ATTN_ROWS), Ti), 8) if not self.no_cluster else 0
        for bi, n0 in enumerate(range(0, N, blk)):
            n1 = min(N, n0 + blk)
            arr = (ctypes.c_void_p * len(_AP))(*[(t[n].data_ptr() + 4 * n0 * row.get(n, 0)) if t[n] is not None else None
                                                for n in _AP])
            if t['XCHG'] is not None:          # own granule region per row block (zero-filled once per pass, see gru256_fwd)
                arr[AP['XCHG']] = t['XCHG'].data_ptr() + 8 * bi * xslots
            tables.append((n1 - n0, arr))
        self._attn_keep = t
        return tables

    def set_regularity(self, overwrought=0.0, oneorder_dynamic=0.0, variance_between_row=0.0, alignment_entropy=0.0):
        """Weights of the optional alignment regularisers (tacotron.py:140-171).  Call before forward()."""
        self.regularity = dict(overwrought=float(overwrought), oneorder_dynamic=float(oneorder_dynamic),
                               variance_between_row=float(variance_between_row), alignment_entropy=float(alignment_entropy))

    @property
    def has_regularity(self):
        return any(v != 0.0 for v in self.regularity.values())

    # ---- loss (models/tacotron.py:127-171) -----------------------------------------------------------------------------
    def loss(self, linear_targets, with_grad=True):
        N, Ti, To, S = self.dims
        Mp, st = N * To, self.st
        self.linear_targets = linear_targets
        tail_done = with_grad and getattr(self, '_tail_done', False)     # linear loss + gradient already taken behind the post-net biGRU
        if not tail_done:
            self.loss_sums = self.dslot(32)             # 2 losses x TACO_L1_REPL (8) replica pairs
        dMEL = self.buf('dmel_loss', Mp, self.nm) if with_grad else None
        dLIN = self.buf('dlin', Mp, self.L.ld_lin) if with_grad else None
        # the (small) mel loss runs beside the linear loss on the second decoder stream
        cur = torch.cuda.current_stream()
        if tail_done:
            lib.taco_l1_loss(self.mel_outputs, self.nm, self.mel_targets, self.nm, dMEL, self.nm, self.loss_sums, Mp, self.nm, 0,
                             1.0 / (Mp * self.nm), 0.0, st)
        else:
            ev = torch.cuda.Event(); ev.record(cur)
            self.stream_b.wait_event(ev)
            with torch.cuda.stream(self.stream_b):
                lib.taco_l1_loss(self.mel_outputs, self.nm, self.mel_targets, self.nm, dMEL, self.nm, self.loss_sums, Mp, self.nm, 0,
                                 1.0 / (Mp * self.nm), 0.0, self.st)
            lib.taco_l1_loss(self.linear_outputs, self.nf, linear_targets, self.nf, dLIN, self.L.ld_lin,
                             self.loss_sums[16:], Mp, self.nf, self.npri, 0.5 / (Mp * self.nf), 0.5 / (Mp * self.npri), st)
            cur.wait_stream(self.stream_b)
        self.reg_sum = None
        if self.has_regularity:
            # loss_regularity (tacotron.py:140-171): value + gradient wrt the alignments, consumed by the attention BPTT
            self.reg_sum = self.dslot(2)
            rg = self.regularity
            lib.taco_align_regularity(self._bufs['ALIGN'], self.buf('dALIGN_reg', N * S, Ti), self.reg_sum, N, S, Ti, rg['overwrought'],
                                      rg['oneorder_dynamic'], rg['variance_between_row'], rg['alignment_entropy'], st)

    def loss_values(self):
        """(loss, mel_loss, linear_loss) as Python floats -- synchronises.  loss includes loss_regularity (tacotron.py:171),
        which is kept in self.loss_regularity."""
        N, Ti, To, S = self.dims
        s = self.loss_sums.cpu().numpy().reshape(2, 8, 2).sum(axis=1)      # [mel | linear] x [all columns, priority columns]
        mel = s[0, 0] / (N * To * self.nm)
        lin = 0.5 * s[1, 0] / (N * To * self.nf) + 0.5 * s[1, 1] / (N * To * self.npri)
        self.loss_regularity = float(self.reg_sum.cpu().numpy()[0]) if self.reg_sum is not None else 0.0
        return mel + lin + self.loss_regularity, mel, lin

    # ---- backward ---------------------------------------------------------------------------------------------------------
    def backward(self):
        L, st, b = self.L, self.st, self._bufs
        N, Ti, To, S = self.dims
        r, nm = self.r, self.nm
        Me, Mp, Ms = N * Ti, N * To, N * S
        E = L.Et + L.Es
        self._mark('loss')
        self._side_active = self.overlap_wgrad          # (the gradient buffer was zero-filled at the start of forward)
        if self.world > 1:
            from . import dp
            if self._exchange is None:
                self._exchange = dp.BucketExchange(self.grads, dp.bucket_ranges(self.L, int(os.environ.get('TACO_DP_BUCKETS', '4'))),
                                                   self.world)
            self._exchange.begin()
        nb = len(self._exchange.ranges) if (self.world > 1 and self._exchange is not None) else 0
        dLIN, POST = b['dlin'], b['post_cbhg/out']
        # linear layer (tacotron.py:101)
        self.gemm_dw(POST, dLIN, self.G('linear/kernel'), Mp, 256, self.nf, ldw=L.ld_lin)
        self.colsum(dLIN, self.G('linear/bias'), Mp, L.ld_lin)
        dPOST = self.buf('dpost', Mp, 256)
        if not getattr(self, '_tail_done', False):
            self.gemm_dx(dLIN, self.P('linear/kernel'), dPOST, Mp, 256, L.ld_lin, ldw=L.ld_lin)
        chunks = self._chunks(N, S, Ti, self.pipe_chunks_bwd, plan_env='TACO_CHUNK_PLAN_BWD')[::-1]
        cur = torch.cuda.current_stream()
        dOUT = self.buf('dout', Ms, nm * r)
        D2 = b['D2']
        dD = self.buf('dD2', Ms, 256)
        # Post-net bank input gradient under the lead-in of the decoder backward: the attention BPTT can only start once GRU2 and GRU1
        # have done their first chunk (~0.23 ms in which 128 CUs idle), so the bank's dX (0.22 ms, the last GEMM of the post-net
        # backward) is computed chunk by chunk, last chunk first, on a stream of its own: the decoder backward starts after the first
        # piece.  dOUT starts as residual + mel-loss gradient and every piece ACCUMULATES into it.
        pipe_post = self.post_pipe and len(chunks) > 1
        if pipe_post:
            dB, dHW0 = self.cbhg_bwd('post_cbhg', self.mel_outputs.view(Mp, nm), dPOST, N, To, nm, 8, (256, nm), None, None,
                                     eager=os.environ.get('TACO_POST_EAGER', '0') == '1', bank_dx_later=True, tail=True)
            lib.taco_add(dHW0, b['dmel_loss'], dOUT, Mp * nm, 0, st)
            evs = torch.cuda.Event(); evs.record(cur)
            sd = self.stream_d
            sd.wait_event(evs)
            piece_done = []
            with torch.cuda.stream(sd):
                for (s0, s1) in chunks:
                    f0, f1 = r * s0, r * s1
                    self._timed('dX GEMM (conv_gemm_nt2)', self._gemm_flops(N * (f1 - f0), nm, 1024, 8, 8),
                                lambda: lib.taco_conv_rows_bwd_data(dB, self.P('post_cbhg/conv_bank/kernel'), dOUT.view(Mp, nm), N, To, f0, f1,
                                                                    nm, 1024, 8, 8, 1024, 128, nm, 1, self.st))
                    e = torch.cuda.Event(); e.record(sd)
                    piece_done.append(e)
            # weight / bias gradient of the output projection read the WHOLE dOUT, which stream_d is still accumulating into: whatever
            # stream ends up running them (the side streams at any release point, or this stream when the weight gradients are not
            # deferred at all: TACO_OVERLAP_WGRAD=0) first waits for the last piece
            if self._side_active:
                # (a stream never waits for its own event: inside a HIP-graph capture that self-edge crashes hipStreamEndCapture)
                self._deferred.append(lambda: [ss.wait_event(piece_done[-1]) for ss in self.side_streams if ss is not sd])
            else:
                cur.wait_event(piece_done[-1])
            self.gemm_dw(D2, dOUT.view(Ms, nm * r), self.G('output_projection/kernel'), Ms, 256, nm * r)
            self.colsum(dOUT.view(Ms, nm * r), self.G('output_projection/bias'), Ms, nm * r)
        else:
            dMELp = self.buf('dmel_post', Mp, nm)
            self.cbhg_bwd('post_cbhg', self.mel_outputs.view(Mp, nm), dPOST, N, To, nm, 8, (256, nm), None, dMELp,
                          eager=os.environ.get('TACO_POST_EAGER', '0') == '1', tail=True)
        if nb >= 2:
            self._bucket_ready(0)                       # post-net + linear
        self._mark('post-net bwd')
        if not pipe_post:
            lib.taco_add(dMELp, b['dmel_loss'], dOUT, Mp * nm, 0, st)
            # output projection
            self.dense_bwd(D2, dOUT, 'output_projection', Ms, 256, nm * r, dx=dD)
        # Chunk-pipelined decoder backward (descending chunks): GRU2 BPTT on the current stream, GRU1 BPTT and the attention
        # BPTT on two more streams; the hoisted input-gradient projections of a chunk run between the stages.
        sb, sc_ = (self.stream_b, self.stream_c) if len(chunks) > 1 else (cur, cur)
        dHC = b['dHC']
        if self.no_cluster or not lib.load().taco_attn_cluster_supported(min(N, self.ATTN_ROWS), Ti):
            for k in ('dQ', 'dKEYS', 'dMEM', 'dVPART'):     # accumulators of the per-step kernels; the cluster path writes them
                lib.taco_zero(b[k], b[k].numel() * 4, st)
        dxp = {g: self.buf('dxp%d' % g, Ms, 768) for g in (1, 2)}
        car = {g: self.buf('gcarry%d' % g, N, 256) for g in (1, 2)}
        xg = {g: b['xchg_g%d' % g] for g in (1, 2)}
        Wp = self.P('concat_projection/kernel')
        flush_at = min(len(chunks) - 1, int(os.environ.get('TACO_FLUSH_AT', '99')))
        early = (os.environ.get('TACO_ATTN_LAST_EARLY', '1') != '0' and len(chunks) > 1 and len(self._attn_ptrs) == 1 and not self.no_cluster
                 and bool(lib.load().taco_attn_cluster_supported(min(N, self.ATTN_ROWS), Ti)))
        if early:
            xchg2 = self.buf('xchg_attn_b', (self._attn_keep['XCHG'].numel() + 1) & ~1, dtype=torch.int64)
            lib.taco_zero(xchg2, xchg2.numel() * 8, st)         # this stream launches the kernel that uses it
        for ci, (s0, s1) in enumerate(chunks):
            if pipe_post:
                cur.wait_event(piece_done[ci])
                self.dense_rows_dx(dOUT.view(Ms, nm * r), self.P('output_projection/kernel'), dD, N, S, s0, s1, 256, nm * r, nm * r, 256, 0)
            R, U, C, RH, Hh = (b['g2_%s' % k] for k in ('r', 'u', 'c', 'rh', 'h'))
            self.gru256_bwd(dD, self.P('decoder_gru_2/whg'), self.P('decoder_gru_2/whc'), R, U, C, Hh, dxp[2], car[2], xg[2],
                            N, S, s0, s1)
            ev = torch.cuda.Event(); ev.record(cur)
            sb.wait_event(ev)
            with torch.cuda.stream(sb):
                self.dense_rows_dx(dxp[2], self.P('decoder_gru_2/wx'), dD, N, S, s0, s1, 256, 768, 768, 256, 1)   # dD1 = dD2 + dxp2.Wx2^T
                R, U, C, RH, Hh = (b['g1_%s' % k] for k in ('r', 'u', 'c', 'rh', 'h'))
                self.gru256_bwd(dD, self.P('decoder_gru_1/whg'), self.P('decoder_gru_1/whc'), R, U, C, Hh, dxp[1], car[1], xg[1],
                                N, S, s0, s1)
                # the projections into the attention BPTT's input stay on THIS stream: queued on the attention stream they would
                # wait behind the previous (longer) attention chunk although their inputs are ready, and sit between two
                # attention chunks on the critical path (57 + 68 + 199 us at C2)
                self.dense_rows_dx(dxp[1], self.P('decoder_gru_1/wx'), dD, N, S, s0, s1, 256, 768, 768, 256, 1)   # dY = dD1 + dxp1.Wx1^T
                self.dense_rows_dx(dD, Wp, dHC, N, S, s0, s1, 512, 256, 256, 512, 0)                               # d[h|ctx] = dY.Wp^T
                ev2 = torch.cuda.Event(); ev2.record(sb)
            # The LAST attention chunk is launched on this stream (GRU2 is done with it), BESIDE the chunk before it: it takes the CUs the
            # GRU kernels have left, runs its prologue and polls for the carries, which the chunk before publishes as granules.  Without
            # this the weight-gradient flood, released behind the last GRU1 chunk, found 128 empty CUs at the boundary between the last
            # two attention launches and the last chunk took 800 us instead of 255 (rocprofv3 trace gpurun_out/r3_tr3).  The flood is
            # held back until every workgroup of that launch has reported itself resident (taco_wait_count).
            if early and ci == len(chunks) - 1:
                cur.wait_event(ev2)
                nb, tab = self._attn_ptrs[0]
                tab2 = (ctypes.c_void_p * len(_AP))(*list(tab))         # the same tensors, an exchange buffer of its own
                tab2[AP['XCHG']] = xchg2.data_ptr()
                self._timed('attention recurrence bwd (attn_cluster_bwd_k)', 2 * self._attn_flops(nb, Ti, s1 - s0),
                            lambda: lib.taco_attn_rnn_bwd_chunk(tab2, self._dims(nb, S, Ti, s0, s1), 2 | 4, self._attn_keep['XCHG'], self.st))
            else:
                sc_.wait_event(ev2)
                with torch.cuda.stream(sc_):
                    for nb, tab in self._attn_ptrs:
                        self._timed('attention recurrence bwd (attn_cluster_bwd_k)', 2 * self._attn_flops(nb, Ti, s1 - s0),
                                    lambda: lib.taco_attn_rnn_bwd_chunk(tab, self._dims(nb, S, Ti, s0, s1),
                                                                        1 if (early and ci == len(chunks) - 2) else 0, None, self.st))
            if ci == flush_at or (ci == len(chunks) - 1 and os.environ.get('TACO_DEV_LOAD')):
                if os.environ.get('TACO_DEV_LOAD'):
                    # developer knob (interference study): a synthetic load takes the place of the weight-gradient flood here
                    # "mode:wgs:lds_bytes:reps:megabytes"; combine with TACO_FLUSH_AT=-1 to move the real flood behind the decoder
                    mode, wgs, ldsb, reps, mb = [int(x) for x in os.environ['TACO_DEV_LOAD'].split(':')]
                    src = b['post_cbhg/bank']
                    evd = torch.cuda.Event(); evd.record(cur)
                    ss = self.side_streams[0]
                    ss.wait_event(evd)
                    dll = lib.load()
                    dll.taco_dev_load.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                  ctypes.c_void_p, ctypes.c_void_p]
                    dll.taco_dev_load(src.data_ptr(), min(mb << 20, src.numel() * 4), reps, mode, wgs, ldsb, self.info.data_ptr() + 12,
                                      ss.cuda_stream)
                if ci == flush_at:
                    # post-net weight gradients fill the CUs the recurrences leave idle; released once the GRU BPTT chunks
                    # (which crowd the first attention chunks) are done.  TACO_FLUSH_AFTER: gru2 = behind the last GRU2 chunk (this
                    # stream), gru1 = behind the last GRU1 chunk and its projections (ev2: the flood no longer shares CUs with the
                    # small projection GEMMs that gate the attention chunks, so those are launched back to back and the flood never
                    # finds the attention CUs empty), attn = behind the launch point of the last attention chunk's predecessor
                    after = os.environ.get('TACO_FLUSH_AFTER', 'gru1')
                    if early and self._side_active and ci == len(chunks) - 1:
                        # (only when the launch the gate waits for has already been ISSUED, above: a gate enqueued earlier --
                        # TACO_FLUSH_AT < last chunk -- would spin while the host is still on its way to that launch, and a first
                        # step's one-off host stalls, module loads and allocations, exceed any sensible bound: seen once as a
                        # timeout of the 50 ms gate in the TACO_FLUSH_AT=0 test)
                        dll = lib.load()
                        ctr = self._attn_keep['XCHG'].data_ptr() + 8 * dll.taco_attn_bwd_resident_slot(min(N, self.ATTN_ROWS), Ti)
                        wgs = dll.taco_attn_bwd_workgroups(min(N, self.ATTN_ROWS))
                        self._deferred.insert(0, lambda: lib.taco_wait_count(ctr, wgs, self.err, self.st))
                    self.flush_side(ready=ev2 if (after == 'gru1' and len(chunks) > 1) else None)
        if len(chunks) > 1:
            cur.wait_stream(sb); cur.wait_stream(sc_)
        if early:
            # the reduction over the steps reads what EVERY chunk stored: the early-resident launch above ends once its predecessor has
            # published the carries, which does not order that predecessor's last stores (other stream, kernel possibly not yet retired)
            nb, tab = self._attn_ptrs[0]
            lib.taco_attn_bwd_reduce(tab, self._dims(nb, S, Ti, 0, S), self.st)
        self._mark('decoder bwd')
        dY = dD
        # weight gradients of the decoder (deferred to the side stream)
        for g in (2, 1):
            sc = 'decoder_gru_%d' % g
            gin = b['D1'] if g == 2 else b['Y']
            self.gemm_dw(gin, dxp[g], self.G(sc + '/wx'), Ms, 256, 768)
            self.colsum(dxp[g], self.G(sc + '/bias'), Ms, 768)
            self.gemm_dw_shift(b['g%d_h' % g], dxp[g], self.G(sc + '/whg'), Ms, S, 256, 512, 256, 768, 512)
            self.gemm_dw(b['g%d_rh' % g], dxp[g][:, 512:], self.G(sc + '/whc'), Ms, 256, 256, ldx=256, lddy=768, ldw=256)
        self.gemm_dw(b['HC'], dY, self.G('concat_projection/kernel'), Ms, 512, 256)
        self.colsum(dY, self.G('concat_projection/bias'), Ms, 256)
        HC, dXPa, dP2, dP1, dQ = b['HC'], b['dXPa'], b['dP2'], b['dP1'], b['dQ']
        self.gemm_dw(b['P2'], dXPa, self.G('attention_gru/wx'), Ms, 128, 768)
        self.colsum(dXPa, self.G('attention_gru/bias'), Ms, 768)
        self.gemm_dw_shift(HC, dXPa, self.G('attention_gru/whg'), Ms, S, 256, 512, 512, 768, 512)
        self.gemm_dw(b['aRH'], dXPa[:, 512:], self.G('attention_gru/whc'), Ms, 256, 256, ldx=256, lddy=768, ldw=256)
        self.gemm_dw(HC, dQ, self.G('attention/query_layer/kernel'), Ms, 256, 256, ldx=512)
        self.colsum(b['dVPART'], self.G('attention/attention_v'), N * Ti, 256)      # rows unused by the per-step path stay 0
        self.gemm_dw(b['P1'], dP2, self.G('decoder_prenet/dense_2/kernel'), Ms, 256, 128)
        self.colsum(dP2, self.G('decoder_prenet/dense_2/bias'), Ms, 128)
        dW1 = self.G('decoder_prenet/dense_1/kernel')
        self.gemm_dw(b['frames'], dP1, dW1, Ms, nm, 256)
        self.gemm_dw_shift(HC[:, 256:], dP1, dW1[nm:], Ms, S, 256, 256, 512, 256, 256)
        self.colsum(dP1, self.G('decoder_prenet/dense_1/bias'), Ms, 256)
        # encoder outputs: values (dMEM) + keys path
        ENC, dKEYS, dENC = b['encoder_cbhg/out'], b['dKEYS'], b['dMEM']
        self.gemm_dw(ENC, dKEYS, self.G('attention/memory_layer/kernel'), Me, 256, 256)
        self.gemm_dx(dKEYS, self.P('attention/memory_layer/kernel'), dENC, Me, 256, 256, acc=1)
        if nb >= 4:
            self._bucket_ready(1)                       # attention + decoder
        dA2 = self.buf('d_enc_p2', Me, 128)
        self.cbhg_bwd('encoder_cbhg', b['enc_p2'], dENC, N, Ti, 128, 16, (128, 128), self.input_lengths, dA2, eager=True,
                      after_proj2=(lambda: self._bucket_ready(2)) if nb >= 4 else None)
        # encoder prenet + embeddings
        lib.taco_relu_bwd(b['enc_p2'], dA2, dA2, Me * 128, st)
        dA1 = self.buf('d_enc_p1', Me, 256)
        self.dense_bwd(b['enc_p1'], dA2, 'prenet/dense_2', Me, 256, 128, dx=dA1)
        lib.taco_relu_bwd(b['enc_p1'], dA1, dA1, Me * 256, st)
        dX0 = self.buf('d_emb', Me, E)
        self.dense_bwd(b['emb'], dA1, 'prenet/dense_1', Me, E, 256, dx=dX0)
        self.gnorm2 = self.dslot(2)
        lib.taco_embed_scatter_bwd(self.inputs, self.identities if L.Es else None, dX0, self.G('embedding'),
                                   self.G('embedding_id') if L.Es else None,
                                   self.gnorm2 if (self.tf_sparse_norm and self.world == 1) else None,
                                   N, Ti, L.Et, L.Es, L.vocab, max(L.id_num, 1), st)
        self._mark('encoder bwd')
        if nb:
            self._bucket_ready(nb - 1)                  # embeddings, encoder prenet, conv bank, proj_1: the tail of backward
        if self._side_active:
            self.flush_side()
            for ss in self.side_streams:
                torch.cuda.current_stream().wait_stream(ss)                # join: all weight gradients are complete
            self._side_active = False
        self._mark('side-stream join')

    # ---- optimizer (models/tacotron.py:174-202) ---------------------------------------------------------------------------
    def optimizer_step(self):
        L, st = self.L, self.st
        if self.tf_sparse_norm and self.world == 1:
            # tf.global_norm: dense tensors + un-deduplicated IndexedSlices rows (already in gnorm2)
            lib.taco_sumsq(self.grads[L.dense_start:], L.total - L.dense_start, self.gnorm2, st)
        else:
            lib.taco_sumsq(self.grads, L.total, self.gnorm2, st)
        # the three launches skip themselves on the device when a cluster hand-off of this step timed out (self.err != 0):
        # garbage gradients never reach the weights; the host raises at its next read-back (check_errors)
        # under data parallelism self.grads holds the SUM over replicas: the 1/world of the average is applied inside the
        # kernel (norm and update), not in a pass of its own
        lib.taco_adam_step(self.params, self.grads, self.m, self.v, L.total, self.gnorm2, self.global_step, self.init_lr,
                           1 if self.decay_lr else 0, self.beta1, self.beta2, 1e-8, 1.0, 1.0 / max(self.world, 1), self.info,
                           self.err, st)
        if L.bn_total:
            lib.taco_bn_ema(self.bn, self.bnbatch, L.bn_total, BN_MOMENTUM, self.global_step, self.err, st)   # UPDATE_OPS (tacotron.py:193) + step counter
        else:
            lib.taco_step_inc(self.global_step, self.err, st)
        self._mark('optimizer')

    def allreduce_grads(self):
        """Data parallel exchange step (net-new): completes the bucketed RCCL all-reduce(sum) of the flat
        gradient that backward() launched bucket by bucket (tacotron_multispeaker_amd/dp.py); the current stream waits for
        the communication, the host does not.  The 1/world factor is applied by the optimizer kernels."""
        if self.world > 1 and self._exchange is not None:
            if self.exposed_events is not None:
                e0 = torch.cuda.Event(enable_timing=True); e0.record()
            self._exchange.finish()
            if self.exposed_events is not None:
                e1 = torch.cuda.Event(enable_timing=True); e1.record()
                self.exposed_events.append((e0, e1))

    def train_step(self, inputs, input_lengths, mel_targets, linear_targets, identities=None):
        if self.main_stream is not None and not torch.cuda.is_current_stream_capturing():
            # run the step on the engine's own (high-priority) stream, ordered after / before the caller's stream
            outer = torch.cuda.current_stream()
            self.main_stream.wait_stream(outer)
            with torch.cuda.stream(self.main_stream):
                self._train_step(inputs, input_lengths, mel_targets, linear_targets, identities)
            outer.wait_stream(self.main_stream)
        else:
            self._train_step(inputs, input_lengths, mel_targets, linear_targets, identities)

    def _train_step(self, inputs, input_lengths, mel_targets, linear_targets, identities=None):
        self.forward(inputs, input_lengths, mel_targets, identities, training=True, linear_targets=linear_targets)
        self.loss(linear_targets)
        self.backward()
        self.allreduce_grads()
        self.optimizer_step()
