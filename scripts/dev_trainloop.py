"""Where the train.py loop (Tacotron.submit_step / collect) loses time against back-to-back engine steps: host time in submit and
collect, GPU time between the `done` events of consecutive steps, with and without the feeder / stager path.  C2 shapes."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hparams as H
from models import create_model
from models.tacotron import GlobalStep
from tacotron_multispeaker_amd import synth

N, Ti, To, r = 32, 128, 640, 5
H.hparams.parse('outputs_per_step=%d,batch_size=%d' % (r, N))
pool = [synth.synth_batch(N, Ti, To, r, seed=10 + i) for i in range(4)]
K = int(os.environ.get('K', '40'))


def run(mode):
    feeder = synth.SyntheticFeeder(pool)
    m = create_model('tacotron', H.hparams)
    if mode == 'static':
        b = pool[0]
        m.initialize(b['inputs'], b['input_lengths'], b['mel_targets'], b['linear_targets'])
    else:
        m.initialize(feeder.inputs, feeder.input_lengths, feeder.mel_targets, feeder.linear_targets)
        feeder.start_in_session(None)
    m.add_loss(); m.add_optimizer(GlobalStep())
    for _ in range(8):
        m.run_step()
    torch.cuda.synchronize()
    time.sleep(2.0)
    for _ in range(3):
        m.run_step()
    sub, col, marks, inflight = [], [], [], []
    t0 = time.perf_counter()
    for i in range(K):
        a = time.perf_counter()
        t = m.submit_step()
        e = torch.cuda.Event(enable_timing=True); e.record(); marks.append(e)
        inflight.append(t)
        b_ = time.perf_counter()
        sub.append(b_ - a)
        if len(inflight) == 2:
            m.collect(inflight.pop(0))
            col.append(time.perf_counter() - b_)
    m.collect(inflight.pop(0))
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / K * 1e3
    gpu = [marks[i].elapsed_time(marks[i + 1]) for i in range(len(marks) - 1)]
    print('%-8s wall %.3f ms/step | GPU done-to-done median %.3f max %.3f | host submit median %.3f max %.3f | collect wait median %.3f'
          % (mode, wall, statistics.median(gpu), max(gpu), statistics.median(sub) * 1e3, max(sub) * 1e3, statistics.median(col) * 1e3), flush=True)
    if mode != 'static':
        st = m._stager
        print('         stager host copy %.3f ms/batch' % (st.host_copy_s / max(st.batches, 1) * 1e3), flush=True)
        feeder.stop(); m.stop()
    e = m.engine
    # back to back on the same engine
    args = synth.batch_to_device(pool[0], e.dev)
    for _ in range(3):
        e.train_step(*args)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K):
        e.train_step(*args)
    torch.cuda.synchronize()
    print('         back-to-back %.3f ms/step   status words %s' % ((time.perf_counter() - t0) / K * 1e3, e.err.cpu().tolist()), flush=True)
    del m


for mode in os.environ.get('MODES', 'static,feeder').split(','):
    run(mode)
