"""A/B timing of the eager C2 training step under environment knobs (each variant in a fresh process: the knobs are read at
launch time by the C library).  usage: python scripts/dev_ab.py NAME=VAL[,NAME=VAL] ...   ('base' = no knob)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, os
sys.path.insert(0, %r)
import torch
from tacotron_multispeaker_amd.engine import Engine
from tacotron_multispeaker_amd import synth
cfg = dict(C2=(32, 128, 640, 5, 0), C5=(16, 200, 800, 2, 460), C4=(32, 64, 480, 5, 460))[os.environ.get('AB_CFG', 'C2')]
N, Ti, To, r, idn = cfg
eng = Engine(r=r, id_num=idn, seed=0)
args = synth.batch_to_device(synth.synth_batch(N, Ti, To, r, seed=1234, id_num=idn), eng.dev)
def step(): eng.train_step(*args)
def fwd(): eng.forward(args[0], args[1], args[2], args[4]); eng.loss(args[3])
def t(fn, n):
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t0) / n * 1e3
for _ in range(5): step()
ts = sorted(t(step, 10) for _ in range(5))
tf = sorted(t(fwd, 10) for _ in range(3))
print('%%-40s step %%.3f ms (min of 5x10)  median %%.3f   fwd %%.3f ms   err %%d' %% (os.environ.get('AB_NAME'), ts[0], ts[2], tf[0], int(eng.err[0].item())), flush=True)
''' % ROOT
for spec in sys.argv[1:]:
    env = dict(os.environ, AB_NAME=spec)
    if spec != 'base':
        for kv in spec.split(','):
            k, v = kv.split('=')
            env[k] = v
    subprocess.run([sys.executable, '-c', CHILD], env=env, check=False)
