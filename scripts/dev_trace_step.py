"""12 eager C2 training steps for rocprofv3 --kernel-trace (timeline.py reads the CSV)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tacotron_multispeaker_amd.engine import Engine
from tacotron_multispeaker_amd import synth
cfg = dict(C2=(32, 128, 640, 5, 0), C5=(16, 200, 800, 2, 460), C4=(32, 64, 480, 5, 460))[os.environ.get('CFG', 'C2')]
N, Ti, To, r, idn = cfg
eng = Engine(r=r, id_num=idn, seed=0)
args = synth.batch_to_device(synth.synth_batch(N, Ti, To, r, seed=1234, id_num=idn), eng.dev)
for _ in range(12):
    eng.train_step(*args)
torch.cuda.synchronize()
print('err', eng.err.cpu().tolist())
