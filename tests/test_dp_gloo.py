"""Data-parallel exchange step on CPU (gloo, world_size 2): the bucketed all-reduce of the flat gradient buffer, launched
bucket by bucket out of band as the engine does during backward (tacotron_multispeaker_amd/dp.py BucketExchange), sums the
replicas' gradients and leaves both ranks bit-identical."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import warnings
    warnings.filterwarnings('ignore')
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from oracle import tacotron_np as onp, tacotron_torch as ot
    from tacotron_multispeaker_amd.params import ParamLayout
    from tacotron_multispeaker_amd import dp
    torch.set_num_threads(2)
    r = 5
    P = onp.init_params(seed=0, r=r)                                  # identical weights on every replica
    b = onp.synth_batch(2, 10, 15, r, seed=1234 + rank)               # rank-specific batch
    ts = ot.TrainState(P, torch.float64, r=r)
    grads = {k: v.numpy() for k, v in ts.forward_backward(b)['grads'].items()}
    L = ParamLayout(r=r)
    flat = torch.zeros(L.total, dtype=torch.float32)
    named = dict(P)
    named.update(grads)
    L.load_named(named, flat, torch.zeros(L.bn_total))
    local = flat.clone()
    buckets = dp.bucket_ranges(L, n_buckets=4)
    assert buckets[0][1] == L.total and buckets[-1][0] == 0           # backward order: post-net/linear first
    assert all(a[0] == b_[1] for a, b_ in zip(buckets[:-1], buckets[1:]))
    assert buckets[1][0] == L.entries['attention/memory_layer/kernel'].offset       # attention + decoder
    assert buckets[2][0] == L.entries['encoder_cbhg/proj_2/kernel'].offset          # encoder proj_2 / highways / biGRU
    # Out-of-band exchange as the engine drives it: the gradient buffer starts at zero (backward zero-fills it), each bucket
    # is all-reduced the moment it has been produced while the later ones do not exist yet; three are launched out of
    # band, finish() launches the tail and waits.
    flat.zero_()
    ex = dp.BucketExchange(flat, buckets, world)
    ex.begin()
    for i in range(3):
        b0, b1 = buckets[i]
        flat[b0:b1] = local[b0:b1]                # "backward" produces bucket i ...
        ex.launch(i)                              # ... and its exchange starts while bucket i+1 is still being computed
    b0, b1 = buckets[3]
    flat[b0:b1] = local[b0:b1]
    ex.finish()
    assert ex.order == [0, 1, 2, 3]
    summed = flat.clone()
    flat.mul_(1.0 / world)                        # the engine applies this factor inside the optimizer kernels
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    expect = sum(gathered) / world
    ok_avg = bool(torch.allclose(flat, expect, rtol=0, atol=1e-7)) and bool(torch.equal(summed, gathered[0] + gathered[1]))
    allf = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(allf, flat)
    ok_same = all(torch.equal(allf[0], t) for t in allf)
    # the blocking one-message form gives the same bits
    again = local.clone()
    dp.allreduce_average(again, world, buckets=dp.bucket_ranges(L, n_buckets=1))
    ok_same = ok_same and bool(torch.equal(again, flat))
    q.put((rank, ok_avg, ok_same, float(flat.abs().sum())))
    dist.destroy_process_group()


def test_bucketed_allreduce_two_ranks_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] and r[2] for r in res), res
    assert res[0][3] == res[1][3] and res[0][3] > 0
