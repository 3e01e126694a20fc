"""Symbol table of the reference's Chinese / mixed front end (reference text/symbols.py:20-23):
symbols2 = ['_' (pad, id 0), '~' (eos, id 1)] + json.load('./datasets/normal.json').

The 7350-entry vocabulary file is a data asset of the reference and is NOT shipped here: place the
reference's datasets/normal.json next to the training run (cwd-relative, like the reference; or point
TACO_VOCAB_JSON at it) and it is used.  Without it, placeholder symbols keep the table size -- the only
property the training step depends on (len(symbols2) = 7352 = embedding rows, reference
models/tacotron.py:40)."""
import json
import os

_pad = '_'
_eos = '~'
NUM_SYMBOLS2 = 7352

_path = os.environ.get('TACO_VOCAB_JSON', './datasets/normal.json')
if os.path.isfile(_path):
    with open(_path, 'r') as f:
        _characters2 = json.load(f)
else:
    _characters2 = ['<sym%d>' % i for i in range(NUM_SYMBOLS2 - 2)]

symbols2 = [_pad, _eos] + list(_characters2)
