import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tacotron_multispeaker_amd.engine import Engine
from tacotron_multispeaker_amd import synth
from tacotron_multispeaker_amd._lib import lib
N, Ti, To, r = 32, 128, 640, 5
os.environ['TACO_OVERLAP_WGRAD'] = '0'          # the BPTT alone on the chip
os.environ['TACO_ATTN_LAST_EARLY'] = '0'        # every chunk in the one exchange buffer whose tail holds the stamps
eng = Engine(r=r, seed=0)
args = synth.batch_to_device(synth.synth_batch(N, Ti, To, r, seed=1234), eng.dev)
for _ in range(2):
    eng.forward(args[0], args[1], args[2]); eng.loss(args[3]); eng.backward(); torch.cuda.synchronize()
bw = lib.load().taco_attn_cluster_bwd_xchg_slots(N, Ti)
st = eng._bufs['xchg_attn'][bw - 16:bw].cpu().numpy().astype(np.float64)
steps = eng._chunks(N, To // r, Ti, eng.pipe_chunks_bwd, plan_env='TACO_CHUNK_PLAN_BWD')[0]
steps = steps[1] - steps[0]                    # the last launch of the pass ran the first chunk in time
names = ['loop top (prefetch regs -> LDS)', 'X1 dctx + barrier', 'X1 da partials + poll', 'barrier', 'X2+3 softmax bwd + dq + publish', 'dq gather + barrier',
         'X4 dq.Wq + dcp publish', 'dcp gather + barrier', 'X5 dcp.Whc + dg publish', 'dg gather + barrier', 'prefetch + X6 dxp.Wx + carry dot + publish',
         'dp2 gather + barrier', 'X7 dp2.W2 + publish', 'dp1 gather + barrier', 'prefetch + X8 dp1.W1c', '-']
tot = st.sum()
for n, v in zip(names, st):
    print('%-44s %8.0f cycles/step  %5.1f%%' % (n, v / steps, 100 * v / max(tot, 1)))
print('total cycles/step', tot / steps, 'steps', steps)
