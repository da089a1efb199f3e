"""Builds the HIP shared library in-tree (plancklens_amd/csrc/libplshts.so) with hipcc for gfx950."""
import os
import shutil
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc')
SOURCES = ['api.hip', 'legendre.hip', 'ringfft.hip', 'elementwise.hip', 'tables.cpp']
HEADERS = ['device_plan.h', 'legendre_math.h', 'plshts_internal.h', 'ringfft.h', 'tproj_device.h', os.path.join('..', '..', 'include', 'plshts.h')]
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


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, libname=None):
    """One object per source (compiled in parallel, only when the source or a header is newer), then one link.
    libname: build under another file name (development variants selected at run time with PLSHTS_LIB; always a full
    rebuild into its own object directory, so that PLSHTS_CXXFLAGS variants do not mix with the default objects)."""
    if libname is None and not force and not needs_build():
        return lib_path()
    from concurrent.futures import ThreadPoolExecutor
    srcs = [f for f in SOURCES if os.path.exists(os.path.join(CSRC, f))]
    objdir = os.path.join(CSRC, 'build' if libname is None else 'build_' + os.path.splitext(libname)[0])
    os.makedirs(objdir, exist_ok=True)
    flags = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-result'] + os.environ.get('PLSHTS_CXXFLAGS', '').split()
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    jobs = []
    for f in srcs:
        obj = os.path.join(objdir, os.path.splitext(f)[0] + '.o')
        if force or libname is not None or _stale(obj, [os.path.join(CSRC, f)] + hdrs):
            jobs.append([hipcc()] + flags + ['-c', f, '-o', obj])
    if verbose:
        for c in jobs:
            print(' '.join(c))
    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
        list(ex.map(lambda c: subprocess.check_call(c, cwd=CSRC), jobs))
    objs = [os.path.join(objdir, os.path.splitext(f)[0] + '.o') for f in srcs]
    out = os.path.join(CSRC, libname or LIBNAME)
    subprocess.check_call([hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs, cwd=CSRC)
    return out


if __name__ == '__main__':
    import sys
    print(build(force=True, verbose=True, libname=sys.argv[1] if len(sys.argv) > 1 else None))
