"""Text -> id sequence for the npy feeder (reference text/__init__.py:51-62, 93-114): every symbol present in
symbols2 maps to its id, unknown symbols are dropped, '{...}' sections are split on whitespace, and the EOS
id (1) is appended."""
import re

from text.symbols import symbols2

_symbol_to_id2 = {s: i for i, s in enumerate(symbols2)}
_id_to_symbol2 = {i: s for i, s in enumerate(symbols2)}
_curly_re = re.compile(r'(.*?)\{(.+?)\}(.*)')
EOS_ID = 1
PAD_ID = 0


def _symbols_to_sequence2(symbols):
    return [_symbol_to_id2[s] for s in symbols if s in _symbol_to_id2]


def text_to_sequence2(text, cleaner_names=None):
    sequence = []
    while len(text):
        m = _curly_re.match(text)
        if not m:
            sequence += _symbols_to_sequence2(text)
            break
        sequence += _symbols_to_sequence2(m.group(1))
        sequence += _symbols_to_sequence2(m.group(2).split())
        text = m.group(3)
    sequence.append(EOS_ID)
    return sequence


def sequence_to_text2(sequence):
    return ''.join(_id_to_symbol2[i] for i in sequence if i in _id_to_symbol2)
