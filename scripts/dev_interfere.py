"""What a concurrently running fp32-MFMA kernel does to latency-bound work on OTHER CUs (round 3 finding: one workgroup running
v_mfma_f32_32x32x2_f32 anywhere on the chip stretches the decoder BPTT 1.6x).  Probes, each alone and beside a background kernel:
  valu   one workgroup, fixed count of dependent packed-fp32 FMAs            -> time ~ 1 / core clock
  xchg   the cluster all-gather micro-benchmark (8-byte granules through L2) -> time ~ L2 / fabric hand-off latency
  lds    (via valu probe with LDS?) not needed
Background (separate stream): mfma = ONE workgroup of dependent fp32 MFMAs, valu = 128 workgroups of packed FMAs, mem = streaming reads."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tacotron_multispeaker_amd._lib import lib
dll = lib.load()
xb = dll.taco_dev_xchg_bench
xb.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 8 + [ctypes.c_void_p]
ld = dll.taco_dev_load
ld.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
x = torch.zeros(1 << 20, dtype=torch.int64, device='cuda'); err = torch.zeros(4, dtype=torch.int32, device='cuda'); sink = torch.zeros(8, device='cuda')
big = torch.zeros(64 << 20, dtype=torch.float32, device='cuda')
s_probe, s_bg = torch.cuda.Stream(), torch.cuda.Stream()


def probe(kind):
    st = s_probe.cuda_stream
    if kind == 'valu':
        return lambda: ld(None, 0, 2000, 2, 1, 1024, sink.data_ptr(), st)                    # 2000 x 256 dependent-ish pk_fma
    if kind == 'mfma':
        return lambda: ld(None, 0, 2000, 1, 1, 1024, sink.data_ptr(), st)
    return lambda: xb(x.data_ptr(), err.data_ptr(), sink.data_ptr(), 16, 8, 64, 1000, 1, 0, 512, 1, st)   # 1000 rounds, same-XCD granules


def background(kind):
    st = s_bg.cuda_stream
    if kind == 'mfma1':
        return lambda: ld(None, 0, 60000, 1, 1, 100000, sink.data_ptr() + 16, st)             # ONE workgroup, ~25 ms
    if kind == 'mfma128':
        return lambda: ld(None, 0, 60000, 1, 128, 100000, sink.data_ptr() + 16, st)
    if kind == 'valu128':
        return lambda: ld(None, 0, 60000, 2, 128, 100000, sink.data_ptr() + 16, st)
    if kind == 'mem':
        return lambda: ld(big.data_ptr(), big.numel() * 4, 60, 0, 512, 48000, sink.data_ptr() + 16, st)
    return None


def run(pk, bk):
    p, b = probe(pk), background(bk)
    for _ in range(2):
        p()
    torch.cuda.synchronize()
    if b is not None:
        b()
        time.sleep(0.002)                       # the background kernel is running
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(s_probe):
        e0.record(); p(); e1.record()
    e1.synchronize()
    t = e0.elapsed_time(e1)
    torch.cuda.synchronize()
    return t


for pk in ('valu', 'mfma', 'xchg'):
    base = run(pk, None)
    print('%-5s alone %.3f ms' % (pk, base), ' | '.join('%s %.3f (x%.2f)' % (bk, t, t / base) for bk, t in
                                                       ((bk, run(pk, bk)) for bk in ('mfma1', 'mfma128', 'valu128', 'mem'))), flush=True)
