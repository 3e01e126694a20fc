"""Hyper-parameter surface of the reference (hparams.py:5-59) without TensorFlow: same 28 names and defaults,
`hparams.parse("a=b,c=d")` with type coercion by the default's type (unknown key -> ValueError, like
tf.contrib.training.HParams), `hparams.values()`, attribute assignment (train.py:55 sets num_GPU) and
`hparams_debug_string()`."""


class HParams(object):
    def __init__(self, **kwargs):
        object.__setattr__(self, '_types', {})
        for k, v in kwargs.items():
            self.add_hparam(k, v)

    def add_hparam(self, name, value):
        self._types[name] = type(value)
        object.__setattr__(self, name, value)

    def __setattr__(self, name, value):
        if name not in self._types:            # train.py:55 / synthesizer.py:20 assign new fields freely
            self._types[name] = type(value)
        object.__setattr__(self, name, value)

    @staticmethod
    def _coerce(name, typ, text):
        text = text.strip()
        if typ is bool:
            low = text.lower()
            if low in ('true', '1'):
                return True
            if low in ('false', '0'):
                return False
            raise ValueError('Could not parse hparam %s: %r is not a bool' % (name, text))
        if typ is int:
            try:
                return int(text)
            except ValueError:
                f = float(text)
                if f != int(f):
                    raise ValueError('Could not parse hparam %s: %r is not an int' % (name, text))
                return int(f)
        if typ is float:
            return float(text)
        return text

    def parse(self, values):
        """'name=value,name=value' overrides (train.py:299, eval.py:81)."""
        if not values:
            return self
        for item in values.split(','):
            if not item.strip():
                continue
            if '=' not in item:
                raise ValueError('Could not parse hparam %r' % item)
            name, text = item.split('=', 1)
            name = name.strip()
            if name not in self._types:
                raise ValueError('Unknown hyperparameter type for %s' % name)
            object.__setattr__(self, name, self._coerce(name, self._types[name], text))
        return self

    def values(self):
        return {k: getattr(self, k) for k in self._types}


# Default hyperparameters (names, order and values of reference hparams.py:5-53):
hparams = HParams(
    cleaners='english_cleaners',
    # Audio:
    num_mels=80,
    num_freq=1025,
    sample_rate=20000,
    frame_length_ms=50,
    frame_shift_ms=12.5,
    preemphasis=0.97,
    min_level_db=-100,
    ref_level_db=20,
    # Model:
    outputs_per_step=1,
    # Training:
    batch_size=32,
    adam_beta1=0.9,
    adam_beta2=0.999,
    initial_learning_rate=0.002,
    decay_learning_rate=True,
    use_phone_input=False,
    per_cen_phone_input=0.0,
    # Eval:
    max_iters=2000,
    griffin_lim_iters=100,
    power=1.5,
    # network settings
    embedding_text_channels=256,
    embedding_id_channels=64,
    # input
    bucket_len=1,
    eos=True,
    # regularity
    overwrought=0.0,
    oneorder_dynamic=0.0,
    variance_between_row=0.0,
    alignment_entropy=0.0,
)


def hparams_debug_string():
    values = hparams.values()
    hp = ['    %s: %s' % (name, values[name]) for name in sorted(values)]
    return 'Hyperparameters:\n' + '\n'.join(hp)
