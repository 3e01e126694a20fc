#!/usr/bin/env python3
"""Headline benchmark: mel-frames/sec of the Tacotron multispeaker TRAINING STEP (forward + backward +
gradient all-reduce + clipped Adam) on N MI355X GPUs of one node, BASELINE.json config "LJSpeech
single-speaker, batch_size=32, r=5" (C2: N=32, T_in=128, T_out=640), synthetic LJSpeech-shaped batches,
random-init weights (TF initialisers), fp32 arithmetic like the reference.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line (rank 0).  Extra objects on that line:
  roofline      -- the dominant MFMA kernel (post-net proj_1 conv as fp32 implicit GEMM), timed live with HIP
                   events on the launch stream: algorithmic FLOPs per launch / average duration vs the fp32
                   MFMA peak (157.3 TFLOP/s, MI355X_MICROARCH.md)
  step_roofline -- the whole step against the same peak: 349.5 GFLOP / ms_per_step
  cpu_baseline  -- the CPU stand-in (oracle/tacotron_torch.py, fp32, same step) timed on this box's host cores
                   (rank 0, N=1 only); TF-1 itself cannot run anywhere in this pipeline (SURVEY.md 8(c,d)).
"""
import argparse
import json
import os
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


def time_dominant_kernel(eng, N, To, iters=20):
    """post_cbhg proj_1: conv1d k=3, 1024 -> 256 over [N*To] rows = the largest single MFMA launch."""
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
    # traffic: HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, raw counter values;
    # profiles/r01_pmc_gemm_v1_v2.md) -- measured offline on the same kernel and shape, not in this run
    return dict(kernel='conv_gemm_nn2<64,64,32,3> (post_cbhg/proj_1 conv1d k=3 1024->256, M=%d)' % M,
                bound='mfma', achieved=flops / (ms * 1e-3) / 1e12, peak=PEAK_FP32_MFMA_TFLOPS, unit='TFLOP/s',
                frac=flops / (ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, traffic=181.2e6 if M == 20480 else None,
                traffic_unit='bytes/launch (160.7 MB FETCH_SIZE + 20.5 MB WRITE_SIZE; algorithmic 108 MB)',
                avg_launch_us=ms * 1e3, flops_per_launch=flops)


def cpu_baseline(N, Ti, To, r, steps=2):
    """CPU stand-in of the same training step (oracle/tacotron_torch.py, fp32) on this box's host cores."""
    import warnings
    warnings.filterwarnings('ignore')
    from oracle import tacotron_np as onp, tacotron_torch as ot
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = min(cores, 16)                 # the GPU box's CPU share per GPU; more threads only add contention
    torch.set_num_threads(cores)
    P = onp.init_params(seed=0, r=r)
    ts = ot.TrainState(P, torch.float32, r=r)
    ts.step(onp.synth_batch(2, 32, 40, r, seed=1))         # warm-up (tiny)
    b = onp.synth_batch(N, Ti, To, r, seed=1234)
    t = time.time()
    for _ in range(steps):
        ts.step(b)
    dt = (time.time() - t) / steps
    return dict(value=N * To / dt, unit='mel-frames/sec', cores=cores, kind='port', sec_per_step=dt,
                sample='%d full training steps of the C2 batch (N=%d,T_in=%d,T_out=%d,r=%d), fp32 PyTorch-CPU restatement '
                       '(oracle/tacotron_torch.py); TF-1 unavailable' % (steps, N, Ti, To, r))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--config', default='C2', choices=['C1', 'C2', 'C4', 'C5', 'C2x2', 'C2x4'])   # C2xK: K times the C2 batch per GPU
    a = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
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

    cfg = dict(C1=(2, 128, 640, 5, 0), C2=(32, 128, 640, 5, 0), C4=(32, 64, 480, 5, 460), C5=(16, 200, 800, 2, 460),
               C2x2=(64, 128, 640, 5, 0), C2x4=(128, 128, 640, 5, 0))[a.config]
    N, Ti, To, r, id_num = cfg
    eng = Engine(r=r, id_num=id_num, seed=0, device=dev)       # identical weights on every replica (RandomState(0))
    eng.world = world
    pool = [synth.batch_to_device(synth.synth_batch(N, Ti, To, r, seed=1234 + rank * 1000 + i, id_num=id_num), dev) for i in range(4)]
    static = [t.clone() if t is not None else None for t in pool[0]]

    def load(i):
        for s, t in zip(static, pool[i % len(pool)]):
            if s is not None:
                s.copy_(t)

    def fwd_bwd():
        eng.forward(static[0], static[1], static[2], static[4])
        eng.loss(static[3])
        eng.backward()

    # eager warm-up allocates the workspace; then capture the step into HIP graphs
    fwd_bwd(); eng.allreduce_grads(); eng.optimizer_step()
    torch.cuda.synchronize()
    use_graph = not a.no_graph
    if use_graph:
        g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1):
            fwd_bwd()
        with torch.cuda.graph(g2):
            eng.optimizer_step()

    def step(i):
        load(i)
        if use_graph:
            g1.replay()
            eng.allreduce_grads()       # RCCL all-reduce of the flat fp32 gradient (no-op at world 1)
            g2.replay()
        else:
            fwd_bwd(); eng.allreduce_grads(); eng.optimizer_step()

    for i in range(a.warmup):
        step(i)

    # HIP-graph replay vs eager launches: keep whichever is faster on this box (the multi-stream graph is not always
    # the winner once the step is GPU-bound); decided on 3 untimed steps each, identically on every rank (max over ranks)
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
        if world > 1:                    # same decision on every rank: compare the slowest rank's times
            import torch.distributed as dist
            tt = torch.tensor([tg, te], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            tg, te = float(tt[0].item()), float(tt[1].item())
        use_graph = tg <= te

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
    barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(a.warmup + i)
    torch.cuda.synchronize(); barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    loss = eng.loss_values()[0]

    if rank == 0:
        ms = dt / a.steps * 1e3
        frames = world * N * To
        fl = 3.0 * step_flops(N, Ti, To, r, 256 + (64 if id_num > 1 else 0))
        out = {
            'metric': 'mel-frames/sec/node (batch32, r=5) training step', 'value': frames / (dt / a.steps),
            'unit': 'mel-frames/sec', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': ms,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': '%s: LJSpeech-shaped batch_size=%d/GPU, T_in=%d, T_out=%d, outputs_per_step=%d, id_num=%d, '
                                   'full training step (fwd+bwd+allreduce+clipped Adam)' % (a.config, N, Ti, To, r, id_num),
                       'global_batch': N * world, 'parallelism': 'dp%d' % world, 'hip_graph': use_graph},
            'loss_after': loss,
            'step_roofline': {'bound': 'mfma', 'achieved': fl / (ms * 1e-3) / 1e12, 'peak': PEAK_FP32_MFMA_TFLOPS,
                              'unit': 'TFLOP/s', 'frac': fl / (ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                              'flops_per_step': fl},
        }
        out['roofline'] = time_dominant_kernel(eng, N, To) if a.config in ('C2',) else None
        if world == 1 and not a.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(N, Ti, To, r)
            out['gpu_over_cpu'] = out['value'] / out['cpu_baseline']['value']
        else:
            out['cpu_baseline'] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
