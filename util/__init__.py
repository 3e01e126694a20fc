"""Small training-loop helpers with the interface train.py expects from the reference's util package
(ValueWindow: util/__init__.py:1-22 there)."""
from collections import deque


class ValueWindow(object):
    """Running statistics over the most recent `window_size` appended values."""

    def __init__(self, window_size=100):
        self._values = deque(maxlen=int(window_size))

    def append(self, x):
        self._values.append(x)

    def reset(self):
        self._values.clear()

    @property
    def count(self):
        return len(self._values)

    @property
    def sum(self):
        return float(sum(self._values)) if self._values else 0

    @property
    def average(self):
        return self.sum / self.count if self._values else 0.0
