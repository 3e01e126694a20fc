import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tacotron_multispeaker_amd.engine import Engine
from tacotron_multispeaker_amd import synth
N, Ti, To, r = 32, 128, 640, 5
eng = Engine(r=r, seed=0)
args = synth.batch_to_device(synth.synth_batch(N, Ti, To, r, seed=1234), eng.dev)
for _ in range(2):
    eng.forward(args[0], args[1], args[2]); torch.cuda.synchronize()
slots = eng._attn_slots(N, Ti)
from tacotron_multispeaker_amd._lib import lib
fw = lib.load().taco_attn_cluster_xchg_slots(N, Ti)
st = eng._bufs['xchg_attn'][fw - 16:fw].cpu().numpy().astype(np.float64)
ch = eng._chunks(N, To // r, Ti)[-1]
S = ch[1] - ch[0]                  # the stamps of the LAST launch (last chunk) survive
names = ['loop/top', 'A p1 dot+publish', 'A gather+barrier', 'B p2 dot+publish', 'B gather+bar', 'C gates dot+publish', 'C cand-x dot+gather+bar',
         'D cand dot+publish', 'D gather+bar', 'E query+bar', 'F scores+publish+gh dot+gather+bar', '-', '-', 'GH softmax+ctx+publish', 'GH gather+bar', '-']
tot = st.sum()
for n, v in zip(names, st):
    print('%-22s %8.0f cycles/step  %5.1f%%' % (n, v / S, 100 * v / max(tot, 1)))
print('total cycles/step', tot / S)
