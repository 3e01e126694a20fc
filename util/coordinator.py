"""Minimal stand-in for tf.train.Coordinator as used by the reference (train.py:60,139,263-266;
datasets/datafeeder_npy.py:88-93): a shared stop flag that remembers the first exception."""
import threading


class Coordinator(object):
    def __init__(self):
        self._stop = threading.Event()
        self._exc = None

    def should_stop(self):
        return self._stop.is_set()

    def request_stop(self, ex=None):
        if ex is not None and self._exc is None:
            self._exc = ex
        self._stop.set()

    @property
    def exception(self):
        return self._exc
