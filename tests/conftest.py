import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


class HostLib:
    """Calls the C ABI with HOST tensors: copies every tensor argument to the device, runs the
    kernel, copies everything back (tests only; lets one test body serve emulator and GPU)."""

    def __init__(self, lib, dev):
        self._lib, self._dev = lib, dev

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if self._dev.type == 'cpu':
            return fn

        def call(*args):
            import torch
            devargs = [a.to(self._dev) if isinstance(a, torch.Tensor) else a for a in args]
            rc = fn(*devargs)
            torch.cuda.synchronize()
            for a, da in zip(args, devargs):
                if isinstance(a, torch.Tensor):
                    a.copy_(da)
            return rc
        return call


class Backend:
    def __init__(self, name, lib, dev, eng):
        self.name, self.lib, self.dev, self.eng = name, lib, dev, eng
        self.hostlib = HostLib(lib, dev)

    def d(self, t):
        return t.to(self.dev)


@pytest.fixture(params=[pytest.param('emu', id='emu'), pytest.param('gpu', id='gpu', marks=pytest.mark.gpu)])
def backend(request):
    """'emu': the kernels compiled for the host and run through the fiber emulator (CPU tests);
    'gpu': libvfs_hip.so on cuda:0 (the parity tests proper, -m gpu)."""
    import torch
    from vfs_amd import engine
    if request.param == 'emu':
        from tests.emu_util import emu_lib
        lib, dev = emu_lib(), torch.device('cpu')
    else:
        from vfs_amd._lib import get_lib
        if not torch.cuda.is_available():
            pytest.skip('needs a GPU (run with -m gpu on the MI355X box)')
        lib, dev = get_lib(), torch.device('cuda:0')   # raises loudly if the .so is missing
    eng = engine.Engine(lib=lib)
    engine.set_shared_engine(eng)
    yield Backend(request.param, lib, dev, eng)
    if dev.type == 'cuda':
        torch.cuda.synchronize()
    engine._ENGINES.clear()


@pytest.fixture()
def emu_backend():
    """CPU-only tests of the kernels (the GPU runs a full-size counterpart of the same properties)"""
    import torch
    from vfs_amd import engine
    from tests.emu_util import emu_lib
    eng = engine.Engine(lib=emu_lib())
    engine.set_shared_engine(eng)
    yield Backend('emu', eng.lib, torch.device('cpu'), eng)
    engine._ENGINES.clear()


@pytest.fixture()
def gpu_backend():
    import torch
    from vfs_amd import engine
    from vfs_amd._lib import get_lib
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU (run with -m gpu on the MI355X box)')
    eng = engine.Engine(lib=get_lib())
    engine.set_shared_engine(eng)
    yield Backend('gpu', eng.lib, torch.device('cuda:0'), eng)
    torch.cuda.synchronize()
    engine._ENGINES.clear()


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(REPO, 'tests', 'golden')
