"""Builds the HIP shared library in-tree (plancklens_amd/csrc/libplshts.so) with hipcc for gfx950."""
import os
import shutil
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc')
SOURCES = ['api.hip', 'legendre.hip', 'ringfft.hip', 'elementwise.hip', 'qe_fused.hip', 'tables.cpp']
HEADERS = ['device_plan.h', 'legendre_math.h', 'plshts_internal.h', 'ringfft.h', os.path.join('..', '..', 'include', 'plshts.h')]
LIBNAME = 'libplshts.so'


def lib_path():
    return os.path.join(CSRC, LIBNAME)


def hipcc():
    exe = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(exe):
        raise RuntimeError('hipcc not found: the HIP library cannot be built')
    return exe


def needs_build():
    so = lib_path()
    if not os.path.exists(so):
        return True
    t = os.path.getmtime(so)
    return any(os.path.exists(os.path.join(CSRC, f)) and os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False, libname=None):
    """libname: build under another file name (development variants selected at run time with PLSHTS_LIB)"""
    if libname is None and not force and not needs_build():
        return lib_path()
    srcs = [f for f in SOURCES if os.path.exists(os.path.join(CSRC, f))]
    cmd = [hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-Wno-unused-result'] + \
          os.environ.get('PLSHTS_CXXFLAGS', '').split() + ['-o', libname or LIBNAME] + srcs
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return os.path.join(CSRC, libname) if libname else lib_path()


if __name__ == '__main__':
    import sys
    print(build(force=True, verbose=True, libname=sys.argv[1] if len(sys.argv) > 1 else None))
