"""The C-ABI library loads without a GPU and exports every entry point declared in include/plshts.h."""
import os
import re

from plancklens_amd import _build, _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_and_exports_header_symbols():
    so = _build.build()
    assert os.path.exists(so)
    L = _lib.lib()
    hdr = open(os.path.join(ROOT, 'include', 'plshts.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    declared = sorted(set(re.findall(r'\b(pl_[a-z0-9_]+)\s*\(', hdr)))
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(L, name), 'symbol %s declared in plshts.h but not exported' % name
    assert set(declared) == set(_lib.SYMBOLS), set(declared) ^ set(_lib.SYMBOLS)
    assert L.pl_version() == 1


def test_no_gpu_is_reported_not_faked():
    """Without a device the product path must fail loudly (no CPU fallback)."""
    L = _lib.lib()
    n = L.pl_device_count()
    if n <= 0:
        import ctypes
        import pytest
        h = ctypes.c_void_p()
        rc = L.pl_plan_create(8, 16, ctypes.byref(h))
        assert rc != 0 and len(L.pl_last_error()) > 0
        from plancklens_amd import shts
        with pytest.raises(RuntimeError):
            shts.get_plan(8, 16)


def test_product_never_imports_oracle():
    """Nothing under plancklens_amd/ may import, call or link the oracle (test infrastructure only)."""
    pkg = os.path.join(ROOT, 'plancklens_amd')
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.cpp', '.h')):
                txt = open(os.path.join(d, f)).read()
                assert 'oracle' not in txt, (d, f)
