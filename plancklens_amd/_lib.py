"""ctypes binding of the C ABI declared in include/plshts.h (the drop-in boundary).

The product path has no CPU fallback: if the HIP library is missing or no GPU is visible, every transform
raises.  Loading the library itself needs no GPU (used by the `not gpu` tests to check the exported symbols).
"""
import ctypes
import os

from . import _build

_LIB = None

SYMBOLS = ['pl_version', 'pl_last_error', 'pl_device_count', 'pl_plan_create', 'pl_plan_create_opts', 'pl_plan_fork', 'pl_plan_destroy', 'pl_plan_npix',
           'pl_plan_nalm', 'pl_plan_bytes', 'pl_plan_side_stream', 'pl_plan_executed_steps', 'pl_plan_useful_steps', 'pl_plan_create_shard', 'pl_phase_pack', 'pl_phase_unpack', 'pl_phase_pack_doubles', 'pl_alm_keep_mgroups', 'pl_map_pack_doubles', 'pl_map_pack_rings', 'pl_map_unpack_rings', 'pl_alm2map', 'pl_alm2map_grad', 'pl_alm2map_pair', 'pl_alm2map_batch2', 'pl_alm2map_grad_pair', 'pl_map2alm', 'pl_map2alm_ind', 'pl_store_addresses', 'pl_plan_phase_doubles', 'pl_legendre_synth', 'pl_legendre_synth_grad',
           'pl_legendre_anal', 'pl_phase2map', 'pl_map2phase', 'pl_almxfl', 'pl_alm2cl', 'pl_alm_copy', 'pl_axpy',
           'pl_alm_dot', 'pl_axpy_dev', 'pl_alm_splice', 'pl_alm_splice_fl', 'pl_cg_dot_axpy', 'pl_almxfl_add', 'pl_alm_lincomb', 'pl_template_project', 'pl_cg_fwd_tt', 'pl_cg_fwd_pp', 'pl_gemv', 'pl_gemv_split', 'pl_gemv_split_dot', 'pl_gemv_split_dot_count', 'pl_alm_splice_dot_b', 'pl_alm_splice_dot_count', 'pl_copy_slim',
           'pl_almxfl_b', 'pl_alm_copy_b', 'pl_alm_splice_b', 'pl_almxfl_add_b', 'pl_alm_dot_b', 'pl_axpy_dev_b', 'pl_cg_dot_axpy_b', 'pl_post_dots_count', 'pl_plan_arm_post_dots', 'pl_cg_axpy_pre_b', 'pl_template_project_b', 'pl_lowrank_update_b',
           'pl_cg_fwd_tt_b', 'pl_cg_fwd_pp_b', 'pl_cg_fwd_pp_qu_b', 'pl_template_project_md_b', 'pl_template_md_scratch_doubles',
           'pl_plan_fft_all_generic', 'pl_cg_fwd_tt_md_b', 'pl_cg_fwd_tt_lr_b', 'pl_gemv_b', 'pl_map_mul', 'pl_map_qu_weight', 'pl_map_cmul', 'pl_qe_lens_product', 'pl_map_add_normal', 'pl_alm_unit_phases', 'pl_fma64_peak_tflops', 'pl_fma64_rate_tflops', 'pl_profile_enable', 'pl_profile_read']

PL_HOST, PL_DEVICE = 0, 1


class PlanOpts(ctypes.Structure):
    """pl_plan_opts of include/plshts.h (-1: the library's default)"""
    _fields_ = [('fft_legacy', ctypes.c_int), ('fft_split_min', ctypes.c_int), ('fft_nyq_min', ctypes.c_int), ('fft_min_fast', ctypes.c_int),
                ('fft_generic_nside', ctypes.c_int), ('seed_tables', ctypes.c_int)]


class PlshtsError(AssertionError):
    """Raised on a non-zero return of the C ABI (AssertionError-compatible: the reference only asserts)."""


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    so = os.environ.get('PLSHTS_LIB', '') or _build.lib_path()  # PLSHTS_LIB: an alternative build of the same library
    if not os.path.exists(so):
        raise RuntimeError('%s is missing: run `python -c "import __graft_entry__ as g; g.build()"` '
                           '(the HIP path has no CPU fallback)' % so)
    L = ctypes.CDLL(so)
    vp, i32, i64, dbl = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_double
    L.pl_version.restype = i32
    L.pl_last_error.restype = ctypes.c_char_p
    L.pl_device_count.restype = i32
    L.pl_plan_create.argtypes = [i32, i32, ctypes.POINTER(vp)]
    L.pl_plan_create_opts.argtypes = [i32, i32, i32, i32, ctypes.POINTER(PlanOpts), ctypes.POINTER(vp)]
    L.pl_plan_fork.argtypes = [vp, ctypes.POINTER(vp)]
    L.pl_plan_create_shard.argtypes = [i32, i32, i32, i32, ctypes.POINTER(vp)]
    L.pl_phase_pack.argtypes = [vp, i32, vp, vp, i32, i32, i32, i32, vp]
    L.pl_phase_unpack.argtypes = [vp, i32, vp, vp, i32, i32, i32, i32, vp]
    L.pl_phase_pack_doubles.argtypes = [vp, i32, i32, i32, i32, i32]
    L.pl_phase_pack_doubles.restype = i64
    L.pl_alm_keep_mgroups.argtypes = [i32, i32, vp, i32, i32, vp]
    L.pl_map_pack_doubles.argtypes = [vp, i32, i32]
    L.pl_map_pack_doubles.restype = i64
    L.pl_map_pack_rings.argtypes = [vp, i32, vp, vp, i32, i32, vp]
    L.pl_map_unpack_rings.argtypes = [vp, i32, vp, vp, i32, i32, vp]
    L.pl_plan_destroy.argtypes = [vp]
    for f in ('pl_plan_npix', 'pl_plan_nalm', 'pl_plan_bytes'):
        getattr(L, f).argtypes = [vp]
        getattr(L, f).restype = i64
    L.pl_plan_side_stream.argtypes = [vp, i32]
    L.pl_plan_side_stream.restype = vp
    L.pl_plan_executed_steps.argtypes = [vp, i32, i32]
    L.pl_plan_executed_steps.restype = i64
    L.pl_plan_useful_steps.argtypes = [vp, i32, i32]
    L.pl_plan_useful_steps.restype = i64
    L.pl_plan_phase_doubles.argtypes = [vp, i32]
    L.pl_plan_phase_doubles.restype = i64
    L.pl_alm2map.argtypes = [vp, i32, vp, vp, vp, i32, vp]
    L.pl_alm2map_grad.argtypes = [vp, i32, vp, vp, vp, i32, vp]
    L.pl_alm2map_pair.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp]
    L.pl_alm2map_batch2.argtypes = [vp, i32, vp, vp, vp, vp, vp]
    L.pl_alm2map_grad_pair.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp]
    L.pl_legendre_synth_grad.argtypes = [vp, i32, vp, vp, vp, vp]
    L.pl_map2alm.argtypes = [vp, i32, vp, vp, vp, i32, vp]
    L.pl_map2alm_ind.argtypes = [vp, i32, vp, vp, vp, vp]
    L.pl_store_addresses.argtypes = [i32, ctypes.POINTER(ctypes.c_uint64), vp, vp]
    L.pl_legendre_synth.argtypes = [vp, i32, vp, vp, vp, vp]
    L.pl_legendre_anal.argtypes = [vp, i32, vp, vp, vp, vp]
    L.pl_phase2map.argtypes = [vp, i32, vp, vp, vp]
    L.pl_map2phase.argtypes = [vp, i32, vp, vp, vp]
    L.pl_almxfl.argtypes = [i32, vp, vp, i32, vp, vp]
    L.pl_alm2cl.argtypes = [i32, vp, vp, vp, vp]
    L.pl_alm_copy.argtypes = [i32, vp, i32, vp, vp]
    L.pl_axpy.argtypes = [i64, dbl, vp, vp, vp, vp]
    L.pl_alm_dot.argtypes = [i32, i32, vp, vp, i32, vp, vp]
    L.pl_axpy_dev.argtypes = [i64, vp, vp, dbl, vp, vp, vp]
    L.pl_alm_splice.argtypes = [i32, vp, i32, vp, i32, vp, vp]
    L.pl_alm_splice_fl.argtypes = [i32, vp, i32, vp, vp, i32, vp, vp]
    L.pl_cg_dot_axpy.argtypes = [i32, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, dbl, vp, vp, dbl, vp, vp]
    L.pl_almxfl_add.argtypes = [i32, vp, vp, vp, i32, vp, vp]
    L.pl_alm_lincomb.argtypes = [i32, i32, vp, vp, vp, vp, vp]
    L.pl_gemv.argtypes = [i32, i32, i64, vp, vp, vp, vp]
    L.pl_gemv_split.argtypes = [i32, i32, i32, i64, vp, vp, vp, vp, vp, vp]
    L.pl_gemv_split_dot.argtypes = [i32, i32, i32, i64, vp, vp, vp, vp, vp, vp, i32, vp, vp]
    L.pl_gemv_split_dot_count.argtypes = [i32, i32, i32]
    L.pl_alm_splice_dot_count.argtypes = [i32]
    L.pl_alm_splice_dot_b.argtypes = [i32, i32, vp, i32, vp, vp, i32, vp, vp, i32, vp, vp]
    L.pl_copy_slim.argtypes = [vp, vp, i64, i32, vp]
    L.pl_template_project.argtypes = [i64, i32, vp, vp, vp, vp, vp, vp]
    L.pl_cg_fwd_tt.argtypes = [vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    L.pl_cg_fwd_pp.argtypes = [vp] * 13
    L.pl_almxfl_b.argtypes = [i32, i32, vp, vp, i32, vp, vp]
    L.pl_alm_copy_b.argtypes = [i32, i32, vp, i32, vp, vp]
    L.pl_alm_splice_b.argtypes = [i32, i32, vp, i32, vp, vp, i32, vp, vp]
    L.pl_almxfl_add_b.argtypes = [i32, i32, vp, vp, vp, i32, vp, vp]
    L.pl_alm_dot_b.argtypes = [i32, i32, i32, vp, vp, i32, vp, vp]
    L.pl_axpy_dev_b.argtypes = [i64, i32, vp, vp, dbl, vp, vp, vp]
    L.pl_cg_dot_axpy_b.argtypes = [i32, i32, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, dbl, vp, vp, dbl, vp, vp]
    L.pl_post_dots_count.argtypes = [vp]
    L.pl_plan_arm_post_dots.argtypes = [vp, i32, vp, vp, i32, vp, vp]
    L.pl_cg_axpy_pre_b.argtypes = [i32, i32, vp, i32, vp, vp, vp, vp, vp, vp, vp, dbl, vp, vp, dbl, vp, i32, vp]
    L.pl_template_project_b.argtypes = [i64, i32, i32, vp, vp, vp, vp, vp, vp]
    L.pl_lowrank_update_b.argtypes = [i64, i32, i32, vp, vp, vp, vp, vp, vp]
    L.pl_cg_fwd_tt_b.argtypes = [vp, i32, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    L.pl_cg_fwd_pp_b.argtypes = [vp, i32] + [vp] * 12
    L.pl_cg_fwd_pp_qu_b.argtypes = [vp, i32] + [vp] * 14
    L.pl_template_project_md_b.argtypes = [vp, i32, vp, vp, vp, vp, vp]
    L.pl_template_md_scratch_doubles.argtypes = [vp, i32]
    L.pl_template_md_scratch_doubles.restype = i64
    L.pl_plan_fft_all_generic.argtypes = [vp]
    L.pl_cg_fwd_tt_md_b.argtypes = [vp, i32] + [vp] * 10
    L.pl_cg_fwd_tt_lr_b.argtypes = [vp, i32, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    L.pl_gemv_b.argtypes = [i32, i32, i64, vp, i32, vp, vp, vp]
    L.pl_map_mul.argtypes = [i64, vp, vp, vp, vp]
    L.pl_map_qu_weight.argtypes = [i64, vp, vp, vp, vp, vp, vp]
    L.pl_map_cmul.argtypes = [i64, vp, vp, dbl, vp, vp, dbl, dbl, vp, vp, i32, vp]
    L.pl_qe_lens_product.argtypes = [i64] + [vp] * 12
    L.pl_map_add_normal.argtypes = [i64, vp, vp, dbl, ctypes.c_uint64, vp]
    L.pl_alm_unit_phases.argtypes = [i32, vp, ctypes.c_uint64, vp]
    L.pl_fma64_rate_tflops.argtypes = [i32, i32, vp]
    L.pl_fma64_rate_tflops.restype = dbl
    L.pl_profile_enable.argtypes = [vp, i32]
    L.pl_profile_read.argtypes = [vp, vp, vp]
    L.pl_fma64_peak_tflops.argtypes = [i32, vp]
    L.pl_fma64_peak_tflops.restype = dbl
    _LIB = L
    return L


def check(rc):
    if rc != 0:
        raise PlshtsError(lib().pl_last_error().decode())


def device_count():
    n = lib().pl_device_count()
    if n < 0:
        raise RuntimeError(lib().pl_last_error().decode())
    return n
