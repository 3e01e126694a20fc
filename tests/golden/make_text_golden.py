"""Generates tests/golden/text_sequences.json with the REFERENCE's text.text_to_sequence2 (text/__init__.py:51-62)
and its vocabulary (datasets/normal.json): a few strings -> id sequences, plus the symbol->id pairs they touch.
Import needs inert stubs for tensorflow / unidecode / inflect (absent here; unused by these functions)."""
import json
import os
import sys
from unittest import mock

REF = '/root/reference'
TEXTS = [u'职工 们 爱 厂', u'abc {zh i2 g ong1} 一二三!', u'', u'~_人',
         u'no-such--symbol 大']


def main():
    for m in ('tensorflow', 'unidecode', 'inflect'):
        sys.modules[m] = mock.MagicMock()
    here = os.path.dirname(os.path.abspath(__file__))
    os.chdir(REF)
    sys.path.insert(0, REF)
    import text as ref
    from text.symbols import symbols2
    out = {'num_symbols2': len(symbols2), 'cases': [], 'symbols': {}}
    for t in TEXTS:
        seq = ref.text_to_sequence2(t, ['basic_cleaners'])
        out['cases'].append({'text': t, 'sequence': seq, 'roundtrip': ref.sequence_to_text2(seq)})
        for i in seq[:-1]:                      # the last id is the appended EOS, not a looked-up symbol
            out['symbols'][symbols2[i]] = i
    with open(os.path.join(here, 'text_sequences.json'), 'w') as f:
        json.dump(out, f, ensure_ascii=True, indent=1)
    print(out['num_symbols2'], [c['sequence'] for c in out['cases']])


if __name__ == '__main__':
    main()
