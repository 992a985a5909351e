"""ctypes binding of libvfs_hip.so (include/vfs_hip.h is the single source of truth: the
prototypes are parsed from it).  There is NO fallback: if the HIP library is missing the
product path raises immediately."""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), 'include', 'vfs_hip.h')
LIB_PATH = os.path.join(_HERE, 'csrc', 'libvfs_hip.so')

_CT = {'int': ctypes.c_int, 'long long': ctypes.c_longlong, 'float': ctypes.c_float,
       'double': ctypes.c_double, 'vfs_stream_t': ctypes.c_void_p}


def parse_header(path=HEADER):
    """-> {name: (restype, [(ctype, argname), ...])} for every prototype in the header."""
    src = open(path).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    protos = {}
    for m in re.finditer(r'\b(int|const char\*)\s+(vfs_\w+)\s*\(([^)]*)\)\s*;', src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        alist = []
        if args and args != 'void':
            for a in args.split(','):
                a = ' '.join(a.split())
                if 'char*' in a.replace(' ', ''):
                    alist.append((ctypes.c_char_p, a.split('*')[-1].strip()))
                elif '*' in a:
                    alist.append((ctypes.c_void_p, a.split('*')[-1].strip()))
                else:
                    typ, an = a.rsplit(' ', 1)
                    alist.append((_CT[typ.replace('const ', '').strip()], an))
        protos[name] = (ctypes.c_char_p if 'char' in ret else ctypes.c_int, alist)
    return protos


class VfsError(RuntimeError):
    pass


class VfsLib:
    """Loaded C-ABI library; attribute access gives checked callables taking torch tensors /
    ints / floats (tensors are passed as their data_ptr; None -> NULL)."""

    def __init__(self, path):
        if not os.path.exists(path):
            raise VfsError(f'{path} not found: build it with `python -m vfs_amd.build` '
                           '(there is no CPU fallback for the HIP path)')
        self.path = path
        self.dll = ctypes.CDLL(path)
        self.protos = parse_header()
        self._fns = {}
        for name, (ret, args) in self.protos.items():
            fn = getattr(self.dll, name)  # AttributeError if the symbol is missing
            fn.restype = ret
            fn.argtypes = [t for t, _ in args]
            self._fns[name] = fn

    def last_error(self):
        return self._fns['vfs_last_error']().decode()

    def __getattr__(self, name):
        fn = self.__dict__.get('_fns', {}).get('vfs_' + name)
        if fn is None:
            raise AttributeError(name)

        def call(*args):
            conv = [a.data_ptr() if hasattr(a, 'data_ptr') else a for a in args]
            rc = fn(*conv)
            if isinstance(rc, int) and rc != 0:
                raise VfsError(f'vfs_{name} failed ({rc}): {self.last_error()}')
            return rc
        self.__dict__[name] = call
        return call


_LIB = None


def get_lib():
    """The product library (gfx950).  Raises VfsError when it has not been built."""
    global _LIB
    if _LIB is None:
        _LIB = VfsLib(LIB_PATH)
    return _LIB


def set_lib(lib):
    """Dependency injection for tests (e.g. the host-emulation build of the same sources)."""
    global _LIB
    _LIB = lib
