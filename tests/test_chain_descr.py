"""filt_cinv.default_chain_descr re-derives the reference's default multigrid chains from a table; this test reads the reference's own
literals -- the `chain_descr = [[...]]` list displays inside cinv_t / cinv_p / cinv_tp.__init__ of /root/reference/plancklens/filt/
filt_cinv.py (:112-116, :236-239, :400-407), evaluated in place with stand-in names -- and compares element by element.  Build
container only (the GPU box has no /root/reference)."""
import ast
import os

import numpy as np
import pytest

REF = '/root/reference/plancklens/filt/filt_cinv.py'


class _cd(object):  # stands in for plancklens.qcinv.cd_solve while the literal is evaluated
    tr_cg = 'tr_cg'

    @staticmethod
    def cache_mem():
        return 'cache_mem'


def _reference_literals(lmax, nside, pcf):
    tree = ast.parse(open(REF).read())
    found = {}
    for cls in [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name in ('cinv_t', 'cinv_p', 'cinv_tp')]:
        for node in ast.walk(cls):
            if isinstance(node, ast.Assign) and len(node.targets) == 1 and getattr(node.targets[0], 'id', None) == 'chain_descr' \
                    and isinstance(node.value, ast.List):
                code = compile(ast.Expression(node.value), REF, 'eval')
                found[cls.name] = eval(code, {'pcf': pcf, 'lmax': lmax, 'nside': nside, 'np': np, 'cd_solve': _cd})
    return found


@pytest.mark.skipif(not os.path.exists(REF), reason='needs /root/reference (build container only)')
@pytest.mark.parametrize('lmax,nside', [(1024, 512), (2048, 2048), (3000, 1024)])
def test_default_chain_descr_equals_the_reference_literals(lmax, nside):
    from plancklens_amd.filt import filt_cinv
    from plancklens_amd.qcinv import cd_solve
    pcf = '/some/dir/dense.pk'
    ref = _reference_literals(lmax, nside, pcf)
    assert set(ref) == {'cinv_t', 'cinv_p', 'cinv_tp'}
    for kind, name in (('t', 'cinv_t'), ('p', 'cinv_p'), ('tp', 'cinv_tp')):
        ours = filt_cinv.default_chain_descr(kind, lmax, nside, pcf)
        assert len(ours) == len(ref[name]), (name, len(ours), len(ref[name]))
        for so, sr in zip(ours, ref[name]):
            assert len(so) == len(sr) == 8
            assert so[0] == sr[0] and so[1] == sr[1], (name, so[:2], sr[:2])  # stage id, descriptor strings (blanks included)
            assert so[2:6] == sr[2:6], (name, so[2:6], sr[2:6])                # lmax, nside, iter_max, eps_min
            assert so[6] is cd_solve.tr_cg and sr[6] == 'tr_cg'
            assert isinstance(so[7], cd_solve.cache_mem) and sr[7] == 'cache_mem'
    # the stages must not share their direction caches
    ours = filt_cinv.default_chain_descr('t', lmax, nside, pcf)
    assert len(set(id(s[7]) for s in ours)) == len(ours)
