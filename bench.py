#!/usr/bin/env python3
"""Headline benchmark: mel-frames/sec of the Tacotron multispeaker TRAINING STEP (forward + backward + gradient
all-reduce overlapped with backward + clipped Adam) on N MI355X GPUs of one node, BASELINE.json config "LJSpeech
single-speaker, batch_size=32, r=5" (C2: N=32, T_in=128, T_out=640), synthetic LJSpeech-shaped batches,
random-init weights (TF initialisers), fp32 arithmetic like the reference.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line (rank 0).  Besides the contract fields:
  ms_per_step_median   median of the K per-step times (HIP events on the main stream); ms_per_step = total / K
  roofline             the kernel family with the largest IN-STEP time (HIP events around every launch of the family on the
                       stream it is launched on, during extra eager steps of the same workload): algorithmic FLOPs per launch /
                       average launch duration vs the fp32-MFMA peak (157.3 TFLOP/s); traffic = HBM bytes per launch from the
                       committed rocprofv3 --pmc passes (profiles/, path in the object) or null
  kernel_families      the same figures for every timed family (the recurrences are latency-bound: their fraction of the
                       MFMA peak is reported as what it is)
  gemm_roofline        the largest single MFMA launch (post-net proj_1 conv) timed back to back in isolation
  gemm_products        which GEMMs multiply as three bf16 MFMAs per fp32 product (TACO_X3) and the step time with exact fp32 products
  step_roofline        the whole step: 349.5 GFLOP / ms_per_step vs the same peak, and 1.56 GB vs 8 TB/s
  parity               max relative error / mean |diff| of mel and linear outputs vs the fp32 CPU restatement on the bench batch
  cpu_baseline         the CPU stand-in (oracle/tacotron_torch.py, fp32, same step) on this box's host cores (rank 0, N=1 only);
                       TF-1 itself cannot run anywhere in this pipeline (SURVEY.md 8(c,d))
  allreduce_exposed_ms (N > 1) time the optimizer's stream waits for the gradient exchange after backward has finished
  train_loop           (N = 1) the loop train.py actually runs -- Tacotron.submit_step()/collect() on feeder handles: numpy batches
                       -> stager thread -> pinned buffers -> device, one 128-byte status read-back per step, one step in flight:
                       train_loop_ms_per_step (same C2 batches), ms_per_step_synced (run_step(): the host waits for every step
                       before it enqueues the next), train_loop_shapes (a cycle of 5 different (T_in, T_out) shapes, beside the
                       back-to-back time of the same cycle).  `value` / ms_per_step remain the back-to-back device-resident loop.

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (torch.distributed.run as a child process,
before anything touches the GPU) and exits with its code.
"""
import argparse
import json
import os
import statistics
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from tacotron_multispeaker_amd.engine import Engine  # noqa: E402
from tacotron_multispeaker_amd import synth  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: 256 CU x 2.4 GHz x 256 FLOP/clk
PEAK_HBM_GBS = 8000.0
TRAFFIC_FILE = os.path.join('profiles', 'r03_pmc_traffic.json')    # {family: {"bytes_per_launch": ..., "source": ...}}
KERNEL_STATS_FILE = os.path.join('profiles', 'r03_step_c2_kernel_stats.csv')   # rocprofv3 --kernel-trace --stats of the same workload
# kernel-name fragment of every timed family in that file
# (family labels are the engine's timing labels and the keys of the committed traffic file; since round 3 the weight-gradient family is two
# grouped kernels per call -- bf16x3 products for the large problems, fp32 products for the rest -- and the dX family holds nt3 launches too)
FAMILY_KERNELS = {'dW GEMM (conv_gemm_tn2_group)': ('conv_gemm_tn3_group', 'conv_gemm_tn2_group'), 'dX GEMM (conv_gemm_nt2)': ('conv_gemm_nt2', 'conv_gemm_nt3'),
                  'fwd GEMM (conv_gemm_nn2)': 'conv_gemm_nn2',
                  'attention recurrence bwd (attn_cluster_bwd_k)': 'attn_cluster_bwd_k', 'attention recurrence fwd (attn_cluster_fwd_k)': 'attn_cluster_fwd_k',
                  'decoder GRU(256) bwd (gru256_cluster_bwd_k)': 'gru256_cluster_bwd_k', 'decoder GRU(256) fwd (gru256_cluster_fwd_k)': 'gru256_cluster_fwd_k',
                  'biGRU(128) bwd (gru128_seq_bwd_k)': 'gru128_seq_bwd_k', 'biGRU(128) fwd (gru128_seq_fwd_k)': 'gru128_seq_fwd_k',
                  'highway x4 bwd (highway4_bwd_k)': 'highway4_bwd_k', 'highway x4 fwd (highway4_fwd_k)': 'highway4_fwd_k'}


def step_flops(N, Ti, To, r, E_in=256):
    """Algorithmic forward FLOPs; a training step is 3x."""
    S = To // r
    f = 2 * N * Ti * (E_in * 256 + 256 * 128) + 2 * N * Ti * 128 * 128 * 136 + 2 * N * Ti * (3 * 2048 * 128 + 3 * 128 * 128)
    f += 2 * N * Ti * 8 * 128 * 128 + 4 * Ti * N * 256 * 384 + 2 * N * Ti * 256 * 256
    f += S * (2 * N * (336 * 256 + 256 * 128) + 2 * N * 384 * 768 + 2 * N * 256 * 256 + 5 * N * Ti * 256 + 2 * N * 512 * 256
              + 4 * N * 512 * 768 + 2 * N * 256 * 80 * r)
    f += 2 * N * To * 80 * 128 * 36 + 2 * N * To * (3 * 1024 * 256 + 3 * 256 * 80) + 2 * N * To * 80 * 128
    f += 2 * N * To * 8 * 128 * 128 + 4 * To * N * 256 * 384 + 2 * N * To * 256 * 1025
    return float(f)


def step_bytes(N, Ti, To, r, n_params):
    """Algorithmic HBM bytes per step: targets + outputs once, 40 B/param, saved activations written + read."""
    S = To // r
    return 2.0 * 4 * N * To * (80 + 1025) + 40.0 * n_params + 2.0 * 4 * (5632 * N * Ti + 5088 * N * S + 4128 * N * To)


def time_isolated_gemm(eng, N, To, iters=20):
    """post_cbhg proj_1: conv1d k=3, 1024 -> 256 over [N*To] rows = the largest single MFMA launch, back to back."""
    from tacotron_multispeaker_amd._lib import lib
    M = N * To
    X = eng._bufs['post_cbhg/pool']
    Y = eng._bufs['post_cbhg/c1']
    W, b = eng.P('post_cbhg/proj_1/kernel'), eng.P('post_cbhg/proj_1/bias')
    st = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        lib.taco_conv_gemm_fwd(X, W, b, Y, M, To, 1024, 256, 3, 0, 1024, 256, 256, 1, 0, st.cuda_stream)
    e0.record(st)
    for _ in range(iters):
        lib.taco_conv_gemm_fwd(X, W, b, Y, M, To, 1024, 256, 3, 0, 1024, 256, 256, 1, 0, st.cuda_stream)
    e1.record(st)
    e1.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flops = 2.0 * M * 3 * 1024 * 256
    tf = flops / (ms * 1e-3) / 1e12
    return dict(kernel='conv_gemm_nn2<64,64,32,3> (post_cbhg/proj_1 conv1d k=3 1024->256, M=%d), isolated back-to-back launches' % M,
                bound='mfma', achieved=tf, peak=PEAK_FP32_MFMA_TFLOPS, unit='TFLOP/s', frac=tf / PEAK_FP32_MFMA_TFLOPS,
                avg_launch_us=ms * 1e3, flops_per_launch=flops)


def kernel_families(eng, run_step, steps=5):
    """In-step duration of every timed kernel family: HIP events around each launch on its own stream (Engine._timed)."""
    eng.ktime = []
    for i in range(steps):
        run_step(i)
    torch.cuda.synchronize()
    fam = {}
    for name, flops, e0, e1 in eng.ktime:
        f = fam.setdefault(name, dict(launches=0, ms=0.0, flops=0.0))
        f['launches'] += 1; f['ms'] += e0.elapsed_time(e1); f['flops'] += flops
    eng.ktime = None
    traffic = {}
    if os.path.exists(os.path.join(ROOT, TRAFFIC_FILE)):
        traffic = json.load(open(os.path.join(ROOT, TRAFFIC_FILE)))
    # rocprofv3's own durations of the same kernels (first wave start -> last wave end), from the committed --kernel-trace --stats
    # summary of the same workload.  The live HIP-event bracket above also contains the time a launch waits on its stream for CUs
    # (a persistent launch needs 128 empty ones) and for the events its stream waits on, so it reads 10-25 % longer for the
    # recurrence chunks; `frac` (contract: measured live) uses the events, `frac_rocprof` the profiler's average duration.
    prof = {}
    if os.path.exists(os.path.join(ROOT, KERNEL_STATS_FILE)):
        import csv
        for r in csv.DictReader(open(os.path.join(ROOT, KERNEL_STATS_FILE))):
            for famname, key in FAMILY_KERNELS.items():
                if any(k in r['Name'] for k in ((key,) if isinstance(key, str) else key)):
                    a = prof.setdefault(famname, [0, 0.0])
                    a[0] += int(r['Calls']); a[1] += float(r['TotalDurationNs'])
    out = []
    for name, f in fam.items():
        tf = f['flops'] / (f['ms'] * 1e-3) / 1e12 if f['ms'] > 0 else 0.0
        latency = 'recurrence' in name or 'GRU' in name
        tr = traffic.get(name, {})
        out.append(dict(kernel=name, bound='latency (serial recurrence; charged against the MFMA peak)' if latency else 'mfma',
                        launches_per_step=f['launches'] / steps, ms_per_step=f['ms'] / steps,
                        avg_launch_us=f['ms'] / f['launches'] * 1e3, flops_per_launch=f['flops'] / f['launches'],
                        achieved=tf, peak=PEAK_FP32_MFMA_TFLOPS, unit='TFLOP/s', frac=tf / PEAK_FP32_MFMA_TFLOPS,
                        # (per timed launch of the family: a weight-gradient call is one or two grouped kernels, so per-step bytes / calls)
                        traffic=(tr['bytes_per_step'] / (f['launches'] / steps) if tr.get('bytes_per_step') else tr.get('bytes_per_launch')),
                        traffic_source=tr.get('source')))
        # (only where the profile holds exactly the launches timed here: the small row-blocked projection GEMMs of the decoder
        # pipeline are launched untimed, so the GEMM families of the profile average over more, shorter launches)
        if name in prof and prof[name][0] and abs(prof[name][0] / 6.0 - f['launches'] / steps) < 0.5:
            us = prof[name][1] / prof[name][0] / 1e3
            # the committed profile has the same launches per step, so FLOPs per launch carry over
            out[-1].update(rocprof_avg_launch_us=us, frac_rocprof=f['flops'] / f['launches'] / (us * 1e-6) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                           rocprof_source=KERNEL_STATS_FILE)
        elif name in prof and prof[name][0] and name.startswith('dW GEMM'):
            # one timed call = one or two grouped kernels: compare per STEP (every weight-gradient launch is timed, so the totals match)
            ms_prof = prof[name][1] / 6.0 / 1e6
            out[-1].update(rocprof_ms_per_step=ms_prof, rocprof_kernel_launches_per_step=prof[name][0] / 6.0,
                           frac_rocprof=f['flops'] / steps / (ms_prof * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, rocprof_source=KERNEL_STATS_FILE)
        if name.startswith('dW GEMM') or name.startswith('dX GEMM'):
            out[-1]['products'] = ('fp32-equivalent FLOPs over the fp32 MFMA peak; the large problems of this family multiply as three bf16 MFMAs per '
                                   'product (gemm_products): per fp32-equivalent FLOP that pipe peaks at 2500 / 3 = 833 TFLOP/s')
    out.sort(key=lambda d: -d['ms_per_step'])
    return out


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(P, N, Ti, To, r, id_num, steps=3):
    """CPU stand-in of the same training step (oracle/tacotron_torch.py, fp32) on this box's host cores."""
    import warnings
    warnings.filterwarnings('ignore')
    from oracle import tacotron_np as onp, tacotron_torch as ot
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count() or 1
    # the GPU box gives one GPU's job a share of 16 CPUs whatever the host's core count: more threads than that only
    # oversubscribe the quota (a 200+-thread run of this step did not finish within 7 minutes)
    cores = min(cores, int(os.environ.get('TACO_CPU_THREADS', '16')))
    torch.set_num_threads(cores)
    ts = ot.TrainState(P, torch.float32, id_num=id_num, r=r)
    ts.step(onp.synth_batch(2, 32, 40, r, seed=1, id_num=id_num))         # warm-up (tiny)
    b = onp.synth_batch(N, Ti, To, r, seed=1234, id_num=id_num)
    ts.step(b)                                                            # warm-up (full size)
    t = time.time()
    for _ in range(steps):
        ts.step(b)
    dt = (time.time() - t) / steps
    return dict(value=N * To / dt, unit='mel-frames/sec', cores=cores, cpu=cpu_model(), kind='port', sec_per_step=dt,
                sample='%d full training steps (after 1 warm-up) of the bench batch (N=%d,T_in=%d,T_out=%d,r=%d), fp32 PyTorch-CPU '
                       'restatement (oracle/tacotron_torch.py) on %d threads (the CPU share of a 1-GPU job on this box; host has %d logical CPUs); TF-1 unavailable' % (steps, N, Ti, To, r, cores, os.cpu_count() or 0))


def parity_vs_oracle(eng, P, batch, r, id_num):
    """mel / linear outputs of the HIP forward vs the fp32 CPU restatement on the same weights and batch."""
    from oracle import tacotron_torch as ot
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    Pt = ot.to_torch(P, torch.float32, requires_grad=False)
    with torch.no_grad():
        ref = ot.forward(Pt, batch['inputs'], batch['input_lengths'], torch.tensor(batch['mel_targets']), batch['identities'],
                         id_num, r)
    dev_b = synth.batch_to_device(batch, eng.dev)
    eng.forward(dev_b[0], dev_b[1], dev_b[2], dev_b[4])
    torch.cuda.synchronize()
    out = {}
    for k, mine in (('mel', eng.mel_outputs), ('linear', eng.linear_outputs)):
        a, c = mine.cpu().numpy().astype(np.float64), ref[k + '_outputs'].numpy().astype(np.float64)
        out[k + '_max_rel'] = float(np.abs(a - c).max() / np.abs(c).max())
        out[k + '_l1'] = float(np.abs(a - c).mean())
    out['reference'] = 'oracle/tacotron_torch.py fp32 forward (the stand-in for the TF reference, parity unpinned: DESIGN.md section 2)'
    return out


def train_loop_legs(model, eng, cfg, steps, warmup):
    """The step loop of train.py (reference train.py:139-152) on synthetic feeder batches; see the module docstring."""
    N, Ti, To, r, id_num = cfg

    def np_pool(shapes, seed0):
        return [synth.synth_batch(N, ti, to, r, seed=seed0 + i, id_num=id_num) for i, (ti, to) in enumerate(shapes)]

    def run(pool, pipelined, K, W):
        feeder = synth.SyntheticFeeder(pool)
        model.attach_feeder(feeder)
        feeder.start_in_session(None)
        inflight, done, t0, frames = [], 0, None, 0
        try:
            while done < W + K:
                if pipelined:
                    while len(inflight) < 2 and done + len(inflight) < W + K:
                        inflight.append(model.submit_step())
                    out = model.collect(inflight.pop(0))
                else:
                    out = model.run_step()
                assert out is not None
                done += 1
                if done == W:
                    t0 = time.perf_counter()
                elif done > W:
                    frames += pool[(done - 1) % len(pool)]['mel_targets'].shape[1] * N
        finally:
            dt = time.perf_counter() - t0
            st = model._stager
            host_copy_ms = st.host_copy_s / max(st.batches, 1) * 1e3
            feeder.stop()
            model.stop()
            torch.cuda.synchronize()
        return dt / K * 1e3, frames / dt, host_copy_ms

    out = {}
    same = np_pool([(Ti, To)] * 4, 4321)
    ms, fps, hc = run(same, True, steps, warmup)
    out['train_loop_ms_per_step'] = ms
    out['train_loop_mel_frames_per_sec'] = fps
    out['train_loop_host_copy_ms_per_batch'] = hc
    ms_s, _, _ = run(same, False, steps, warmup)
    out['ms_per_step_synced'] = ms_s
    # five shapes a length-bucketed LJSpeech feeder would hand over in a row (T_out multiples of r)
    rr = lambda x: max(r, int(round(x / r)) * r)
    shapes = [(Ti, To), (max(8, Ti * 7 // 8), rr(To * 0.875)), (max(8, Ti * 3 // 4), rr(To * 0.75)), (max(8, Ti * 15 // 16), rr(To * 0.95)),
              (max(8, Ti * 5 // 8), rr(To * 0.66))]
    pool = np_pool(shapes, 8765)
    ms_v, fps_v, _ = run(pool, True, steps, max(warmup, 2 * len(shapes)))
    # the same cycle back to back on device-resident batches (no feeder, no read-back)
    dev_pool = [synth.batch_to_device(b, eng.dev) for b in pool]
    for i in range(2 * len(pool)):
        eng.train_step(*dev_pool[i % len(pool)])
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(steps):
        eng.train_step(*dev_pool[i % len(pool)])
    torch.cuda.synchronize()
    b2b = (time.perf_counter() - t) / steps * 1e3
    out['train_loop_shapes'] = dict(shapes=shapes, ms_per_step=ms_v, mel_frames_per_sec=fps_v, back_to_back_ms_per_step=b2b)
    out['train_loop_note'] = ('train.py loop: feeder handles -> stager thread -> pinned ring -> device ring, Tacotron.submit_step/collect, '
                              'one 128-byte status copy + one event wait per step, one step in flight; synced = run_step() every step')
    return out


def dp_machinery_leg(eng, pool, steps):
    """What the data-parallel event graph costs on ONE GPU: the same eager steps with the engine told world = 2 over a ONE-rank RCCL
    communicator (no byte travels, all-reduce over one rank is the identity): four buckets launched out of band on the communication
    stream behind their producers, the optimizer waiting for them, no GRU(256) isolation pad.  Not a scaling number -- RCCL over xGMI
    with real peers is unmeasured in this pipeline -- but the part of the N > 1 step that can be measured here."""
    import socket
    import torch.distributed as dist
    if dist.is_initialized():
        return None
    s_ = socket.socket(); s_.bind(('127.0.0.1', 0)); port = s_.getsockname()[1]; s_.close()
    try:
        dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % port, rank=0, world_size=1, device_id=eng.dev)
    except Exception as e:       # noqa: BLE001
        return {'error': 'one-rank RCCL communicator unavailable: %r' % (e,)}
    try:
        eng.world = 2
        eng.exposed_events = []

        def one(i):
            b = pool[i % len(pool)]
            eng.train_step(b[0], b[1], b[2], b[3], b[4])
        for i in range(5):
            one(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            one(i)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        ex = float(np.mean([e0.elapsed_time(e1) for e0, e1 in eng.exposed_events[-steps:]]))
        order = list(eng._exchange.order)
    finally:
        eng.world = 1
        eng.exposed_events = None
        eng._exchange = None
        dist.destroy_process_group()
    return {'ms_per_step': ms, 'allreduce_wait_ms': ex, 'bucket_order': order,
            'note': 'world assumed 2 over a one-rank RCCL communicator on one GPU: event graph, comm stream, no GRU(256) pad; no bytes travel'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--no-families', action='store_true', help='skip the extra instrumented steps (rocprofv3 runs)')
    ap.add_argument('--config', default='C2', choices=['C1', 'C2', 'C2x', 'C4', 'C5', 'C2x2', 'C2x4'])   # C2xK: K times the C2 batch per GPU
    ap.add_argument('--no-train-loop', action='store_true', help='skip the train.py-loop legs')
    a = ap.parse_args()

    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # started as plain `python bench.py --gpus N`: launch the N ranks as a CHILD process before anything touches the GPU
        # (never an exec) and leave with its exit code
        import subprocess
        port = os.environ.get('MASTER_PORT', '29531')
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(a.gpus), '--master-addr', '127.0.0.1',
               '--master-port', port, os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if a.gpus != world:
        sys.exit('bench.py: --gpus %d but WORLD_SIZE=%d: launch one rank per GPU (python -m torch.distributed.run --nproc-per-node %d ...)'
                 % (a.gpus, world, a.gpus))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(local)
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local))
    else:
        torch.cuda.set_device(0)
    dev = torch.device('cuda', local if world > 1 else 0)

    cfg = dict(C1=(2, 128, 640, 5, 0), C2=(32, 128, 640, 5, 0), C2x=(32, 192, 810, 5, 0), C4=(32, 64, 480, 5, 460),
               C5=(16, 200, 800, 2, 460), C2x2=(64, 128, 640, 5, 0), C2x4=(128, 128, 640, 5, 0))[a.config]
    N, Ti, To, r, id_num = cfg
    from tacotron_multispeaker_amd.params import init_named, ParamLayout
    P = init_named(ParamLayout(id_num=id_num, r=r), 0)          # identical weights on every replica (RandomState(0))
    # the engine is built through the reference's construction API on feeder handles (the train-loop legs drive it that way); the
    # headline loop below calls the same engine directly on device-resident batches
    import hparams as H
    from models import create_model
    from models.tacotron import GlobalStep
    H.hparams.parse('outputs_per_step=%d,batch_size=%d' % (r, N))
    handles = synth.SyntheticFeeder([])
    model = create_model('tacotron', H.hparams)
    model.initialize(handles.inputs, handles.input_lengths, handles.mel_targets, handles.linear_targets, identities=handles.identities,
                     id_num=id_num, named_params=P, device=dev)
    model.add_loss()
    model.add_optimizer(GlobalStep())
    eng = model.engine
    eng.world = world
    pool = [synth.batch_to_device(synth.synth_batch(N, Ti, To, r, seed=1234 + rank * 1000 + i, id_num=id_num), dev) for i in range(4)]
    static = [t.clone() if t is not None else None for t in pool[0]]

    def load(i):
        for s, t in zip(static, pool[i % len(pool)]):
            if s is not None:
                s.copy_(t)

    def fwd_bwd(b=None):
        b = static if b is None else b
        eng.forward(b[0], b[1], b[2], b[4], linear_targets=b[3])     # as Engine.train_step does
        eng.loss(b[3])
        eng.backward()

    def eager_step(i):
        # eager launches read the pool batch where it lies in HBM; only the HIP-graph replay needs the static input buffer
        # (and pays a 91 MB device-to-device copy per step for it)
        fwd_bwd(pool[i % len(pool)]); eng.allreduce_grads(); eng.optimizer_step()

    # eager warm-up allocates the workspace; then (single GPU) capture the step into HIP graphs.  With more than one rank the
    # step stays eager: the bucket all-reduces are launched from inside backward on the communication stream.
    t_start = time.perf_counter()
    eager_step(0)
    torch.cuda.synchronize()
    use_graph = (not a.no_graph) and world == 1
    if use_graph:
        g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1):
            fwd_bwd()
        with torch.cuda.graph(g2):
            eng.optimizer_step()

    def step(i):
        if use_graph:
            load(i)
            g1.replay()
            g2.replay()
        else:
            eager_step(i)

    for i in range(a.warmup):
        step(i)

    # HIP-graph replay vs eager launches: keep whichever is faster on this box (the multi-stream graph is not always the
    # winner once the step is GPU-bound); decided on 3 untimed steps each
    if use_graph:
        def probe(g):
            nonlocal use_graph
            use_graph = g
            torch.cuda.synchronize(); t = time.perf_counter()
            for i in range(3):
                step(i)
            torch.cuda.synchronize()
            return time.perf_counter() - t
        tg, te = probe(True), probe(False)
        use_graph = tg <= te

    # A one-off host stall of 40-50 ms shows up 1-3 s into a process's GPU life on these boxes (any script, any step count: the first
    # launches after it wait; scripts/dev_firststep.py) -- with K = 20 it would cost 2 ms/step.  Extra untimed steps until the
    # process has been submitting for SETTLE seconds keep it out of the timed region.
    settle = float(os.environ.get('TACO_BENCH_SETTLE', '2.5'))
    i_extra = 0
    if world > 1:
        # every rank must run the SAME number of steps (each contains the gradient exchange): rank 0's clock decides, in rounds of 25
        import torch.distributed as dist
        while True:
            more = torch.tensor([1 if (time.perf_counter() - t_start < settle) else 0], device=dev, dtype=torch.int32)
            dist.broadcast(more, src=0)
            if int(more.item()) == 0:
                break
            for _ in range(25):
                step(i_extra); i_extra += 1
    else:
        while time.perf_counter() - t_start < settle:
            step(i_extra); i_extra += 1
    torch.cuda.synchronize()

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
    if world > 1:
        eng.exposed_events = []
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    # timing events are created lazily at their first record(): a burst of first records can stall the host for tens of ms (seen as
    # one 45 ms step at the start of the timed region in ~half of the runs) -- record every mark once before the clock starts
    for m in marks:
        m.record()
    if os.environ.get('TACO_BENCH_ABSORB', '1') != '0':
        # the first STEP submitted after a device synchronisation pays for the runtime's clean-up of everything submitted before that
        # synchronisation (10-90 ms, growing with the number of steps before it; trivial launches do not trigger it): synchronise once
        # early and let one more untimed step absorb it, so that the timed region's first step only cleans up after that one step
        torch.cuda.synchronize()
        step(0)                                   # one more untimed step: it pays for the clean-up of the warm-up's commands
    barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(a.steps):
        step(a.warmup + i)
        marks[i + 1].record()
    torch.cuda.synchronize(); barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    per_step = [marks[i].elapsed_time(marks[i + 1]) for i in range(a.steps)]
    loss = eng.loss_values()[0]
    errw = [int(x) for x in eng.err.cpu().tolist()]
    err = errw[0]
    exposed = None
    if world > 1:
        exposed = float(np.mean([e0.elapsed_time(e1) for e0, e1 in eng.exposed_events[-a.steps:]]))
        eng.exposed_events = None

    # the instrumented steps contain the gradient exchange: EVERY rank runs them (a collective issued by rank 0 alone would hang)
    fams = kernel_families(eng, eager_step) if not a.no_families else None
    barrier()
    if rank == 0:
        ms = dt / a.steps * 1e3
        frames = world * N * To
        fl = 3.0 * step_flops(N, Ti, To, r, 256 + (64 if id_num > 1 else 0))
        by = step_bytes(N, Ti, To, r, eng.L.total)
        out = {
            'metric': 'mel-frames/sec/node (batch32, r=5) training step', 'value': frames / (dt / a.steps),
            'unit': 'mel-frames/sec', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': ms,
            'ms_per_step_median': statistics.median(per_step), 'ms_per_step_max': max(per_step),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': '%s: LJSpeech-shaped batch_size=%d/GPU, T_in=%d, T_out=%d, outputs_per_step=%d, id_num=%d, '
                                   'full training step (fwd+bwd+allreduce+clipped Adam)' % (a.config, N, Ti, To, r, id_num),
                       'global_batch': N * world, 'parallelism': 'dp%d' % world, 'hip_graph': use_graph},
            'loss_after': loss, 'cluster_handoff_timeouts': err,
            # placement of the persistent clusters over every launch so far: clusters whose members were NOT on one XCD keep the
            # agent-scope granule form (slower, same results)
            'cluster_xcd_fallbacks': errw[1], 'cluster_placements_checked': errw[2],
            'step_roofline': {'bound': 'mfma', 'achieved': fl / (ms * 1e-3) / 1e12, 'peak': PEAK_FP32_MFMA_TFLOPS,
                              'unit': 'TFLOP/s', 'frac': fl / (ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                              'flops_per_step': fl, 'hbm': {'achieved': by / (ms * 1e-3) / 1e9, 'peak': PEAK_HBM_GBS,
                                                            'unit': 'GB/s', 'frac': by / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                                                            'algorithmic_bytes_per_step': by}},
        }
        if exposed is not None:
            out['allreduce_exposed_ms'] = exposed
        if err:
            out['error'] = 'a persistent cluster kernel reported a hand-off timeout: the timed steps are INVALID'
        if fams is not None:
            out['roofline'] = dict(fams[0], timing='HIP events around every launch of the family on its stream, 5 eager steps of the '
                                                   'same workload after the timed region; frac_rocprof: same FLOPs over the average '
                                                   'duration of the kernel in the committed rocprofv3 summary (the event bracket also holds '
                                                   'the wait for CUs and for the stream\'s event dependencies)')
            out['kernel_families'] = fams[1:]
        else:
            out['roofline'] = None
        out['gemm_roofline'] = time_isolated_gemm(eng, N, To) if a.config in ('C2',) else None
        if world == 1 and not a.no_cpu_baseline:
            batch = synth.synth_batch(N, Ti, To, r, seed=1234, id_num=id_num)
            out['parity'] = parity_vs_oracle(eng, eng.export_named('params'), batch, r, id_num)     # the weights after the timed steps
            out['cpu_baseline'] = cpu_baseline(P, N, Ti, To, r, id_num)
            out['gpu_over_cpu'] = out['value'] / out['cpu_baseline']['value']
        else:
            out['cpu_baseline'] = None
        if world == 1:
            # the large GEMMs of the backward pass multiply on the BF16 matrix pipe, three MFMAs per fp32 product (TACO_X3, gemm.hip);
            # the same eager steps with exact fp32 products everywhere, and with the forward GEMMs on the BF16 pipe as well
            def mode_leg(mode):
                prev = os.environ.get('TACO_X3')
                os.environ['TACO_X3'] = mode
                try:
                    best = None
                    for rep in range(2):               # (the first steps after the CPU legs above run into the host's one-off stalls: min of two)
                        for i in range(10):
                            eager_step(i)
                        torch.cuda.synchronize()
                        eager_step(0)                  # absorbs the clean-up after the synchronisation, as in the timed region
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        for i in range(a.steps):
                            eager_step(i)
                        e1.record(); e1.synchronize()
                        t = e0.elapsed_time(e1) / a.steps
                        best = t if best is None else min(best, t)
                    return best
                finally:
                    if prev is None:
                        os.environ.pop('TACO_X3', None)
                    else:
                        os.environ['TACO_X3'] = prev
            out['gemm_products'] = {
                'mode': os.environ.get('TACO_X3', '1'),
                'note': 'TACO_X3: 0 = exact fp32 products (v_mfma_f32_32x32x2_f32) everywhere; 1 (default, the timed steps) = input- and '
                        'weight-gradient GEMMs of the large layers as a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on v_mfma_f32_32x32x16_bf16 with fp32 '
                        'accumulation (error ~4e-6 of the result norm against float64; forward pass and every recurrence exact fp32); '
                        '2 = the forward GEMMs too',
                'ms_per_step_exact_fp32_products': mode_leg('0'),
                'ms_per_step_forward_too': mode_leg('2')}
        if world == 1 and not a.no_train_loop:
            out.update(train_loop_legs(model, eng, cfg, a.steps, a.warmup))
            out['dp_machinery'] = dp_machinery_leg(eng, pool, a.steps)
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()              # rank 0 is still timing / printing: nobody tears the communicator down under it
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
