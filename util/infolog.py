"""Console + run-log writer with the interface of the reference's util/infolog.py:8-20 (`init(path)`, `log(msg)`):
every message is echoed to stdout and, once init() was called, appended to the log file with a millisecond
timestamp.  Each init() starts a new banner-separated section in the (appended) file."""
import atexit
import datetime
import threading

_BANNER = '-' * 65
_lock = threading.Lock()
_sink = None


def _stamp():
    now = datetime.datetime.now()
    return '%s.%03d' % (now.strftime('%Y-%m-%d %H:%M:%S'), now.microsecond // 1000)


def _close_logfile():
    global _sink
    with _lock:
        if _sink is not None:
            _sink.close()
        _sink = None


def init(filename):
    global _sink
    _close_logfile()
    with _lock:
        _sink = open(filename, 'a', encoding='utf-8')
        _sink.write('\n%s\nStarting new training run\n%s\n' % (_BANNER, _BANNER))
        _sink.flush()


def log(msg):
    print(msg, flush=True)
    with _lock:
        if _sink is not None:
            _sink.write('[%s]  %s\n' % (_stamp(), msg))
            _sink.flush()


atexit.register(_close_logfile)
