from .tacotron import Tacotron


def create_model(name, hparams):
    """Factory of the reference (models/__init__.py:4-8)."""
    if name == 'tacotron':
        return Tacotron(hparams)
    else:
        raise Exception('Unknown model: ' + name)
