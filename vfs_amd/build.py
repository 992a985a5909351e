"""Build libvfs_hip.so (gfx950) in-tree with hipcc.  `python -m vfs_amd.build [--emu]`.

--emu builds tests/emu/_build/libvfs_emu.so instead: the SAME sources compiled for the host
against the fiber emulator in tests/emu (test infrastructure; never loaded by the product path).
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(CSRC, 'libvfs_hip.so')
EMU_DIR = os.path.join(REPO, 'tests', 'emu')
EMU_LIB = os.path.join(EMU_DIR, '_build', 'libvfs_emu.so')


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')))


def _stale(target, extra=()):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    deps = _sources() + glob.glob(os.path.join(CSRC, '*.h')) + glob.glob(os.path.join(REPO, 'include', '*.h'))
    deps += list(extra)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile_all(cmd_for, objdir):
    os.makedirs(objdir, exist_ok=True)
    objs, procs = [], []
    for src in _sources():
        obj = os.path.join(objdir, os.path.basename(src) + '.o')
        objs.append(obj)
        procs.append((src, subprocess.Popen(cmd_for(src, obj), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode:
            raise RuntimeError(f'compile failed on {src}:\n{out.decode()}')
        if out.strip():
            print(out.decode())
    return objs


def build_hip(force=False):
    if not force and not _stale(LIB):
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objs = _compile_all(lambda src, obj: [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall',
                                          '-Wno-unused-function', '-c', src, '-o', obj],
                        os.path.join(CSRC, 'build'))
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs)
    return LIB


def build_emu(force=False):
    hdr = glob.glob(os.path.join(EMU_DIR, 'hip', '*.h'))
    if not force and not _stale(EMU_LIB, hdr):
        return EMU_LIB
    cxx = os.environ.get('EMU_CXX', '/opt/rocm/lib/llvm/bin/clang++')
    objs = _compile_all(lambda src, obj: [cxx, '-x', 'c++', '-std=c++17', '-O2', '-fPIC', '-I', EMU_DIR,
                                          '-Wno-unknown-attributes', '-Wno-unused-value', '-c', src, '-o', obj],
                        os.path.dirname(EMU_LIB))
    subprocess.check_call([cxx, '-shared', '-fPIC', '-o', EMU_LIB] + objs + ['-lpthread'])
    return EMU_LIB


if __name__ == '__main__':
    if '--emu' in sys.argv:
        print(build_emu(force='--force' in sys.argv))
    else:
        print(build_hip(force='--force' in sys.argv))
