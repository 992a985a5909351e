"""ctypes binding of libvfs_hip.so (include/vfs_hip.h is the single source of truth: the
prototypes are parsed from it).  There is NO fallback: if the HIP library is missing the
product path raises immediately."""
import ctypes
import os
import sys
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), 'include', 'vfs_hip.h')
TUNING_HEADER = os.path.join(os.path.dirname(_HERE), 'include', 'vfs_hip_tuning.h')      # vfs_set_option: the A/B switchboard, not part of the operator contract
LIB_PATH = os.path.join(_HERE, 'csrc', 'libvfs_hip.so')

_CT = {'int': ctypes.c_int, 'long long': ctypes.c_longlong, 'float': ctypes.c_float,
       'double': ctypes.c_double, 'vfs_stream_t': ctypes.c_void_p}


def parse_header(path=None):
    """-> {name: (restype, [(ctype, argname), ...])} for every prototype in the header(s)."""
    src = open(path).read() if path else open(HEADER).read() + open(TUNING_HEADER).read()
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

    def stream_index(self, name):
        """position of the `vfs_stream_t stream` argument of entry point `name` (without the vfs_ prefix), None if it has none"""
        args = self.protos.get('vfs_' + name, (None, []))[1]
        return next((i for i, (_, an) in enumerate(args) if an == 'stream'), None)

    def cfunc(self, name):
        """the raw ctypes function of entry point `name` (without the vfs_ prefix), or None"""
        return self._fns.get('vfs_' + name)

    def check(self, name, rc):
        if isinstance(rc, int) and rc != 0:
            raise VfsError(f'vfs_{name} failed ({rc}): {self.last_error()}')

    def __getattr__(self, name):
        fn = self.__dict__.get('_fns', {}).get('vfs_' + name)
        if fn is None:
            raise AttributeError(name)

        trace = os.environ.get('VFS_TRACE_CALLS') == '1'      # diagnostics: name every entry point before it runs (with
                                                              # AMD_SERIALIZE_KERNEL=3 the last line names a faulting launch)

        def call(*args):
            conv = [a.data_ptr() if hasattr(a, 'data_ptr') else a for a in args]
            if trace:
                print(f'[vfs] {name}({", ".join(str(c) for c in conv[-14:])})', file=sys.stderr, flush=True)
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
        _LIB = VfsLib(os.environ.get('VFS_HIP_LIB') or LIB_PATH)     # VFS_HIP_LIB: A/B a variant build of the same ABI
    return _LIB


def set_lib(lib):
    """Dependency injection for tests (e.g. the host-emulation build of the same sources)."""
    global _LIB
    _LIB = lib


class Tape:
    """Host-side command list of one launch chain: the C-ABI calls (ctypes function + already converted arguments)
    and the few Python-side actions between them (stream waits, collectives), recorded while a chain runs eagerly
    and replayed by a tight loop afterwards.  Every pointer argument refers to a persistent engine buffer, exactly
    the precondition of the hipGraph capture; unlike a graph the replay can contain RCCL collectives, which is
    what the multi-GPU step needs (there the Python + ctypes marshalling of ~570 launches per step, 25 us each,
    was slower than the GPU)."""

    def __init__(self, lib):
        self.lib = lib
        self.ops = []
        self.meta = {}          # op index -> (family, algorithmic FLOP, algorithmic bytes) of the launches recorded through Engine.timed
        self.next_meta = None

    host_seconds = 0.0      # diagnostics: host time spent replaying tapes (all tapes of the process)
    slowest = None
    timing = None           # bench.py: a list -> every C-ABI launch of a replay is bracketed by events ON ITS OWN STREAM (the
                            # `vfs_stream_t stream` argument of its prototype) and (family, flop, e0, e1, bytes) is appended: per-kernel
                            # durations of the schedule that is actually timed (two streams, overlapping), not of an eager stand-in
    _streams = {}
    event_pool = None       # optional list of pre-created timing events (creating one costs more than recording it)

    def _replay_timed(self):
        import torch
        check = self.lib.check
        out = Tape.timing
        for i, (name, fn, args) in enumerate(self.ops):
            if name is None:
                fn(*args)
                continue
            si = self.lib.stream_index(name)      # from the header's prototype (advisor r05: not "the last argument if it looks like one")
            sp = args[si] if si is not None else None
            st = Tape._streams.get(sp)
            if st is None:
                st = torch.cuda.ExternalStream(sp) if sp else torch.cuda.default_stream()
                Tape._streams[sp] = st
            pool = Tape.event_pool
            if pool:
                e0, e1 = pool.pop(), pool.pop()
            else:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            rc = fn(*args)
            e1.record(st)
            kind, flops, nbytes = self.meta.get(i, ('other:' + name, 0.0, 0.0))
            out.append((kind, flops, e0, e1, nbytes))
            if rc:
                check(name, rc)

    def replay(self):
        import time
        check = self.lib.check
        t0 = time.perf_counter()
        if Tape.timing is not None:
            self._replay_timed()
        elif Tape.slowest is not None:          # VFS_TAPE_PROFILE=1: per-op host time, to find a call that blocks
            for name, fn, args in self.ops:
                t1 = time.perf_counter()
                rc = fn(*args)
                d = time.perf_counter() - t1
                key = name or getattr(fn, '__qualname__', repr(fn))
                a = Tape.slowest.setdefault(key, [0.0, 0])
                a[0] += d
                a[1] += 1
                if name is not None and rc:
                    check(name, rc)
        else:
            for name, fn, args in self.ops:
                rc = fn(*args)
                if name is not None and rc:
                    check(name, rc)
        Tape.host_seconds += time.perf_counter() - t0


class TapeLib:
    """stands in for a VfsLib while a chain is recorded: executes every call and appends it to the tape"""

    def __init__(self, lib, tape):
        self._lib, self._tape = lib, tape
        self.path = lib.path

    def last_error(self):
        return self._lib.last_error()

    def __getattr__(self, name):
        fn = self._lib.cfunc(name)
        if fn is None:
            raise AttributeError(name)
        lib, ops = self._lib, self._tape.ops

        def call(*args):
            conv = tuple(a.data_ptr() if hasattr(a, 'data_ptr') else a for a in args)
            ops.append((name, fn, conv))
            tape = self._tape
            if tape.next_meta is not None:
                tape.meta[len(ops) - 1], tape.next_meta = tape.next_meta, None
            rc = fn(*conv)
            lib.check(name, rc)
            return rc
        self.__dict__[name] = call
        return call
