#!/usr/bin/env python3
"""Two REAL data-parallel ranks through the engine's event graph on ONE MI355X (gpurun box: one GPU).

RCCL refuses two ranks on one device, gloo does not: both ranks run the full training step of Engine.train_step() on cuda:0
(TACO_ALLOW_SHARED_GPU=1, a batch small enough that both ranks' persistent clusters are co-resident), with world = 2, so
backward launches the four gradient buckets out of band on the communication stream behind the producer events
(Engine._bucket_ready -> dp.BucketExchange.launch) and the optimizer applies 1/world inside its kernels.  The all-reduce itself
travels over gloo -- on the device buffer when this torch build's gloo takes device tensors, otherwise staged through pinned
host memory by the exchange subclass below (this script only; the product path is RCCL) -- so what is exercised is everything
around the collective: bucket order, the stream / event wiring, rank-specific batches, the averaged update.

Checks (every rank, 3 steps, a new batch shape every step): bucket launch order [0, 1, 2, 3]; error word 0; the update of step 1
equals the float64 oracle's clip + Adam on the AVERAGE of the two ranks' oracle gradients (dense global norm);
the replicas' parameters and Adam slots are bit-identical after every step; the batch-norm moving statistics differ (per-replica
statistics, independent towers: reference train.py:101-111).

The parent process never touches the GPU: it only starts the two rank processes and collects their logs.

    python scripts/dp_two_ranks.py [--log profiles/r03_dp_two_ranks.log]
"""
import argparse
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parent(a):
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, TACO_ALLOW_SHARED_GPU='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    extra = ['--rccl'] if a.rccl else []
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), '--rank', str(r), '--port', str(port)] + extra, env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs, rc = [], 0
    deadline = time.time() + a.timeout
    for p in procs:
        try:
            out, _ = p.communicate(timeout=max(1.0, deadline - time.time()))
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
            out += '\n[parent] rank killed at the time limit\n'
        outs.append(out)
        rc = rc or p.returncode
    text = ''.join('---- rank %d (exit %s) ----\n%s\n' % (i, procs[i].returncode, o) for i, o in enumerate(outs))
    text += 'RESULT: %s\n' % ('OK' if rc == 0 else 'FAILED')
    print(text, flush=True)
    if a.log:
        os.makedirs(os.path.dirname(os.path.abspath(a.log)), exist_ok=True)
        with open(a.log, 'w') as f:
            f.write(text)
    return 0 if rc == 0 else 1


def train_mode(a):
    """train.py itself with two ranks on the one GPU (gloo transport): toy dataset, 8 steps, checkpoints every 4.  Exercises the
    per-step agreement of the replicas (before a step is enqueued, after it is collected), rank-0 checkpoints behind a barrier and
    the pipelined submit / collect loop under data parallelism."""
    import json
    import tempfile
    import numpy as np
    tmp = tempfile.mkdtemp(prefix='taco_dp_train_')
    rng = np.random.RandomState(0)
    lines = []
    for i in range(16):
        T = 23 + 3 * (i % 5)
        paths = []
        for kind, shape in (('spec', (T, 1025)), ('mel', (T, 80)), ('wav', (T * 250,))):
            p = os.path.join(tmp, '%s-%d.npy' % (kind, i))
            np.save(p, rng.rand(*shape).astype(np.float32))
            paths.append(p)
        lines.append(repr(paths + ['{%s}' % ' '.join('<sym%d>' % rng.randint(0, 7000) for _ in range(4 + i % 7)), i % 3]))
    meta = os.path.join(tmp, 'toy_id_num_3.txt')
    open(meta, 'w', encoding='utf-8').write('\n'.join(lines) + '\n')
    json.dump({'TOY': meta}, open(os.path.join(tmp, 'train_npy_data_dict.json'), 'w'))
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, TACO_ALLOW_SHARED_GPU='1', TACO_DIST_BACKEND='gloo', RANK=str(r), WORLD_SIZE='2', LOCAL_RANK='0',
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), PYTHONPATH=ROOT)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, 'train.py'), '--base_dir', os.path.join(tmp, 'logs'), '--train_data', 'TOY',
                                       '--description', 'toy', '--hparams', 'batch_size=4,outputs_per_step=5,decay_learning_rate=false,initial_learning_rate=0.001',
                                       '--max_steps', '8', '--checkpoint_interval', '4', '--summary_interval', '4'],
                                      env=env, cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs, rc = [], 0
    for p in procs:
        try:
            out, _ = p.communicate(timeout=a.timeout)
        except subprocess.TimeoutExpired:
            p.kill(); out, _ = p.communicate(); out += '\n[parent] rank killed at the time limit\n'
        outs.append(out); rc = rc or p.returncode
    logf = os.path.join(tmp, 'logs', 'logs-tacotron-toy', 'train.log')
    log = open(logf).read() if os.path.exists(logf) else ''
    steps = [int(l.split('Step')[1].split('[')[0]) for l in log.splitlines() if 'avg_sec/step' in l]
    ck = sorted(f for f in os.listdir(os.path.dirname(logf))) if log else []
    ok = (rc == 0 and sorted(steps) == sorted(list(range(1, 9)) * 2) and log.count('Saving checkpoint to:') == 2
          and 'Exiting due to exception' not in log and 'model.ckpt-4' in ck and 'model.ckpt-8' in ck and log.count('Summary at step') == 4)
    text = ''.join('---- rank %d (exit %s), last lines ----\n%s\n' % (i, procs[i].returncode, '\n'.join(o.splitlines()[-6:])) for i, o in enumerate(outs))
    text += 'steps logged by the two ranks: %s\ncheckpoints: %s\nTRAIN RESULT: %s\n' % (sorted(steps), ck, 'OK' if ok else 'FAILED')
    print(text, flush=True)
    if a.log:
        with open(a.log, 'a') as f:
            f.write(text)
    return 0 if ok else 1


def child(a):
    sys.path.insert(0, ROOT)
    import warnings
    warnings.filterwarnings('ignore')
    import numpy as np
    import torch
    import torch.distributed as dist
    rank, world = a.rank, 2
    dev_index = rank if a.rccl else 0
    if a.rccl:          # two DEVICES, the product transport (RCCL over xGMI); needs a box with >= 2 GPUs
        torch.cuda.set_device(dev_index)
        dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % a.port, rank=rank, world_size=world,
                                device_id=torch.device('cuda', dev_index))
    else:
        dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % a.port, rank=rank, world_size=world)
        torch.cuda.set_device(0)
    from oracle import tacotron_np as onp, tacotron_torch as ot          # checker only
    from tacotron_multispeaker_amd import dp
    from tacotron_multispeaker_amd.engine import Engine

    say = lambda *x: print('[rank %d]' % rank, *x, flush=True)
    r, idn = 5, 3
    shapes = [(6, 40, 120), (6, 33, 100), (5, 48, 140)]                     # (N, T_in, T_out): S = 24, 20, 28 -> chunk pipeline on
    P = onp.init_params(seed=0, r=r, id_num=idn)                            # identical weights on every replica
    batches = [[onp.synth_batch(N, Ti, To, r, seed=1234 + 1000 * k + i, id_num=idn) for i, (N, Ti, To) in enumerate(shapes)]
               for k in range(world)]                                       # batches[rank][step]

    dev_ok = True
    try:
        probe = torch.ones(8, device='cuda:%d' % dev_index)
        dist.all_reduce(probe)
        torch.cuda.synchronize()
        dev_ok = bool(probe[0].item() == world)
    except Exception as e:                                                  # this gloo build does not take device tensors
        dev_ok = False
        say('gloo on device tensors unavailable (%s): staging buckets through pinned host memory' % type(e).__name__)
    oks = [torch.tensor([1 if dev_ok else 0], device='cuda:%d' % dev_index if a.rccl else 'cpu')]
    dist.all_reduce(oks[0], op=dist.ReduceOp.MIN)
    dev_ok = bool(oks[0].item())

    class HostStagedExchange(dp.BucketExchange):
        """Same interface and launch points as dp.BucketExchange; the bucket travels host-side (script only)."""

        def launch(self, i):
            if self.world <= 1 or i in self.works:
                return
            b0, b1 = self.ranges[i]
            cs = torch.cuda.current_stream()           # the engine's communication stream, already waiting for the producers
            cs.synchronize()
            host = self.flat[b0:b1].cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            self.flat[b0:b1].copy_(host)               # on the communication stream
            ev = torch.cuda.Event(); ev.record(cs)

            class W:
                def wait(self_):
                    torch.cuda.current_stream().wait_event(ev)
            self.works[i] = W()
            self.order.append(i)

    eng = Engine(r=r, id_num=idn, named_params=P, device='cuda:%d' % dev_index)
    eng.world = world
    if not dev_ok:
        eng._exchange = HostStagedExchange(eng.grads, dp.bucket_ranges(eng.L, 4), world)
    assert eng._gru256_pad(6) == 0, 'the GRU(256) isolation pad must be off under data parallelism'
    assert a.rccl or os.environ.get('TACO_ALLOW_SHARED_GPU') == '1'

    def to_dev(b):
        t = lambda k, dt: torch.tensor(b[k], device=eng.dev, dtype=dt)
        return (t('inputs', torch.int32), t('input_lengths', torch.int32), t('mel_targets', torch.float32),
                t('linear_targets', torch.float32), t('identities', torch.int32))

    # oracle of step 1: both ranks' float64 gradients, averaged, dense global norm, clip + Adam; own batch-norm statistics
    torch.set_num_threads(4)
    ts = ot.TrainState(P, torch.float64, id_num=idn, r=r, tf_sparse_norm=False)
    lasts = [ts.forward_backward(batches[k][0]) for k in range(world)]
    avg = {k: sum(l['grads'][k] for l in lasts) / world for k in lasts[0]['grads']}
    info = ts.apply(dict(grads=avg, sparse_sumsq={}, out=lasts[rank]['out']))      # batch-norm statistics of THIS rank's batch

    ok = True
    for step, b in enumerate(batches[rank]):
        eng.train_step(*to_dev(b))
        torch.cuda.synchronize()
        words = [int(x) for x in eng.err.cpu().tolist()]
        order = list(eng._exchange.order)
        loss = eng.loss_values()[0]
        say('step %d  N=%d T_in=%d T_out=%d  loss %.6f  bucket order %s  err words %s  global norm %.6f'
            % (step + 1, b['inputs'].shape[0], b['inputs'].shape[1], b['mel_targets'].shape[1], loss, order, words,
               float(eng.info[0].item())))
        ok = ok and order == [0, 1, 2, 3] and words[0] == 0
        if step == 0:
            pn = eng.export_named('params')
            worst = max((float(np.abs(pn[k] - v.detach().numpy()).max()), k) for k, v in ts.P.items() if v.requires_grad)
            gn = abs(float(eng.info[0].item()) - info['global_norm']) / info['global_norm']
            say('step 1 vs float64 oracle on the AVERAGED gradient: max |param diff| %.2e (%s), global-norm rel diff %.2e'
                % (worst[0], worst[1], gn))
            ok = ok and worst[0] < 1e-5 and gn < 1e-4
        same = True
        for name in ('params', 'm', 'v'):
            mine_t = getattr(eng, name) if a.rccl else getattr(eng, name).cpu()
            got = [torch.zeros_like(mine_t) for _ in range(world)]
            dist.all_gather(got, mine_t)
            same = same and all(torch.equal(got[0], t) for t in got)
        bn = eng.bn if a.rccl else eng.bn.cpu()
        gb = [torch.zeros_like(bn) for _ in range(world)]
        dist.all_gather(gb, bn)
        say('step %d  replicas bit-identical (params, m, v): %s   BN moving stats differ per replica: %s'
            % (step + 1, same, not torch.equal(gb[0], gb[1])))
        ok = ok and same and not torch.equal(gb[0], gb[1])
        ok = ok and int(eng.global_step.item()) == step + 1
    say('collective transport: %s' % ('RCCL, one device per rank' if a.rccl else 'gloo on the device gradient buffer' if dev_ok
                                      else 'gloo via pinned host staging (script only)'))
    say('PASS' if ok else 'FAIL')
    dist.barrier()
    dist.destroy_process_group()
    return 0 if ok else 1


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--rank', type=int, default=-1)
    ap.add_argument('--port', type=int, default=0)
    ap.add_argument('--log', default='')
    ap.add_argument('--timeout', type=float, default=540.0)
    ap.add_argument('--train', action='store_true', help='run train.py with two ranks instead of the engine-level check')
    ap.add_argument('--rccl', action='store_true', help='two devices and the nccl (RCCL) backend instead of two ranks on one device over gloo')
    a = ap.parse_args()
    if a.rank >= 0:
        sys.exit(child(a))
    sys.exit(train_mode(a) if a.train else parent(a))
