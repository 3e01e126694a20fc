"""ctypes binding of the C-ABI in include/taco_hip.h (the header is authoritative: prototypes are parsed
from it, so Python can never drift from the exported signatures).  Fails loudly when the HIP library is
missing -- there is no CPU fallback in the product path."""
import ctypes
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(HERE), 'include', 'taco_hip.h')
LIBPATH = os.path.join(HERE, 'lib', 'libtaco_hip.so')

_CT = {'int': ctypes.c_int, 'long': ctypes.c_long, 'float': ctypes.c_float, 'double': ctypes.c_double,
       'hipStream_t': ctypes.c_void_p, 'size_t': ctypes.c_size_t}


def parse_header(path=HEADER):
    """-> {name: [ctypes argtypes]} for every `int taco_*(...)` prototype in the header."""
    txt = open(path).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    txt = re.sub(r'//[^\n]*', '', txt)
    protos = {}
    for m in re.finditer(r'\bint\s+(taco_\w+)\s*\(([^)]*)\)\s*;', txt, flags=re.S):
        name, args = m.group(1), m.group(2)
        types = []
        for a in args.split(','):
            a = a.strip()
            if not a or a == 'void':
                continue
            if '*' in a:
                types.append(ctypes.c_void_p)
            else:
                base = a.replace('const', '').split()
                types.append(_CT[base[0]])
        protos[name] = types
    return protos


class _Lib:
    def __init__(self):
        self._dll = None
        self._protos = None

    def load(self):
        if self._dll is not None:
            return self._dll
        if not os.path.exists(LIBPATH):
            raise RuntimeError(
                'tacotron_multispeaker_amd: HIP library %s is missing. Build it with '
                '`python -m tacotron_multispeaker_amd.build` (hipcc --offload-arch=gfx950). '
                'There is no CPU fallback.' % LIBPATH)
        self._dll = ctypes.CDLL(LIBPATH)
        self._protos = parse_header()
        for name, types in self._protos.items():
            fn = getattr(self._dll, name)      # AttributeError if the header declares a missing symbol
            fn.argtypes = types
            fn.restype = ctypes.c_int
        return self._dll

    def __getattr__(self, name):
        dll = self.load()
        fn = getattr(dll, name)
        def call(*args):
            conv = []
            for a in args:
                if a is None:
                    conv.append(None)
                elif hasattr(a, 'data_ptr'):
                    conv.append(a.data_ptr())
                else:
                    conv.append(a)
            rc = fn(*conv)
            if rc != 0:
                raise RuntimeError('%s failed with code %d' % (name, rc))
        call.__name__ = name
        self.__dict__[name] = call
        return call


lib = _Lib()


def stream():
    import torch
    return torch.cuda.current_stream().cuda_stream
