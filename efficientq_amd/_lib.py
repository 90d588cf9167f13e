"""ctypes binding of the C-ABI library ``libeffq_hip.so`` (include/effq_hip.h).

The product path has no CPU fallback: if the library is missing or a call
returns a non-zero status this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

# torch bundles its own HIP runtime (libamdhip64.so.7).  It must be mapped BEFORE libeffq_hip.so so
# that both share ONE runtime instance (same SONAME => the loader reuses it); loading ours first
# would pull /opt/rocm's copy and leave two runtimes in the process ("no ROCm-capable device").
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libeffq_hip.so")

EFFQ_OK = 0
_ERR_NAMES = {1: "EFFQ_ERR_ARG", 2: "EFFQ_ERR_HIP", 3: "EFFQ_ERR_WORKSPACE", 4: "EFFQ_ERR_NO_DEVICE",
              5: "EFFQ_ERR_NOT_CONVERGED"}


class EffqError(RuntimeError):
    pass


class Geom(C.Structure):
    """effq_geom of include/effq_hip.h."""
    _fields_ = [(n, C.c_int32) for n in
                ("N", "C1", "C2", "D", "H", "W", "KD", "KH", "KW", "SD", "SH", "SW", "PD", "PH", "PW")]

    def out_dims(self):
        return ((self.D + 2 * self.PD - self.KD) // self.SD + 1,
                (self.H + 2 * self.PH - self.KH) // self.SH + 1,
                (self.W + 2 * self.PW - self.KW) // self.SW + 1)


class FpState(C.Structure):
    """effq_fp_state of include/effq_hip.h (40 bytes)."""
    _fields_ = [("alpha", C.c_double), ("alpha_prev", C.c_double), ("sums", C.c_double * 2),
                ("iters", C.c_int32), ("done", C.c_int32)]


FP_STATE_BYTES = C.sizeof(FpState)

_P, _SZ, _I, _F, _D, _LL = C.c_void_p, C.c_size_t, C.c_int, C.c_float, C.c_double, C.c_longlong
_GP = C.POINTER(Geom)


class AdmmRunArgs(C.Structure):
    """effq_admm_run_args of include/effq_hip.h."""
    _fields_ = [("A0", C.c_void_p), ("B0", C.c_void_p), ("W0", C.c_void_p), ("b0", C.c_void_p),
                ("c2", C.c_int32), ("n", C.c_int32), ("has_bias", C.c_int32), ("w_levels", C.c_int32),
                ("iters", C.c_int32), ("rho_period", C.c_int32),
                ("rho", C.c_double), ("rho_max", C.c_double), ("eta", C.c_double), ("tol", C.c_double),
                ("geom", Geom), ("loss_kind", C.c_int32), ("act_levels", C.c_int32),
                ("xq", C.c_void_p), ("xidx", C.c_void_p), ("y_fp", C.c_void_p), ("act_alpha_dev", C.c_void_p),
                ("dual", C.c_void_p), ("wstar", C.c_void_p), ("v", C.c_void_p),
                ("G_ring", C.c_void_p), ("Gq_ring", C.c_void_p), ("b_ring", C.c_void_p), ("state_ring", C.c_void_p),
                ("hist", C.c_void_p), ("err_flag", C.c_void_p),
                ("ainv_pool", C.c_void_p), ("n_ainv", C.c_int32),
                ("prox_ws", C.c_void_p), ("prox_ws_bytes", C.c_size_t),
                ("red_ws", C.c_void_p),
                ("fp_ws", C.c_void_p), ("fp_ws_bytes", C.c_size_t),
                ("fp_pred", C.c_void_p), ("fp_traj_ws", C.c_void_p), ("fp_traj_ws_bytes", C.c_size_t),
                ("inv_ws", C.c_void_p), ("inv_ws_bytes", C.c_size_t),
                ("inv_ws_side", C.c_void_p), ("inv_ws_side_bytes", C.c_size_t),
                ("conv_ws", C.c_void_p), ("conv_ws_bytes", C.c_size_t),
                ("stream_main", C.c_void_p), ("stream_loss", C.c_void_p), ("stream_side", C.c_void_p),
                ("stream_side2", C.c_void_p), ("inv_ws_side2", C.c_void_p), ("inv_ws_side2_bytes", C.c_size_t),
                ("loss_Au", C.c_void_p), ("loss_Bu", C.c_void_p), ("loss_syy", C.c_void_p),
                ("loss_planes", C.c_void_p), ("loss_nplanes", C.c_int32), ("res_ring", C.c_void_p)]


class ProfRecord(C.Structure):
    """effq_prof_record of include/effq_hip.h."""
    _fields_ = [("kind", C.c_int32), ("iter", C.c_int32), ("loss_kind", C.c_int32), ("c2", C.c_int32), ("n", C.c_int32),
                ("geom", Geom), ("ms", C.c_float)]


# name -> (restype, argtypes).  Must list every symbol include/effq_hip.h declares.
SIGNATURES = {
    "effq_last_error": (C.c_char_p, []),
    "effq_version": (_I, []),
    "effq_device_count": (_I, [C.POINTER(C.c_int)]),
    "effq_quant_dequant_f32": (_I, [_P, _P, _F, _F, _I, _P, _P, _SZ, _P]),
    "effq_quant_dequant_f64path": (_I, [_P, _P, _D, _D, _I, _P, _P, _P, _SZ, _P]),
    "effq_reduce_ws_bytes": (_SZ, []),
    "effq_abs_sum_f64": (_I, [_P, _SZ, _P, _P, _P]),
    "effq_moments_f64": (_I, [_P, _SZ, _P, _P, _P]),
    "effq_alpha_stats_f64": (_I, [_P, _P, _D, _D, _I, _SZ, _P, _P, _P, _P]),
    "effq_fp_init": (_I, [_P, _P, _P]),
    "effq_fp_update": (_I, [_P, _D, _I, _P]),
    "effq_alpha_fixed_point": (_I, [_P, _SZ, _I, _D, _D, _D, _I, _I, _P, _P, _P]),
    "effq_fp_small_max": (_SZ, []),
    "effq_fixed_point_small": (_I, [_P, _P, _P, _SZ, _I, _D, _D, _D, _I, _P, _P]),
    "effq_fp_coop_max": (_SZ, []),
    "effq_fp_coop_set_spin_limit": (_I, [C.c_uint]),
    "effq_fixed_point_coop": (_I, [_P, _P, _P, _SZ, _I, _D, _D, _D, _I, _P, _P, _P]),
    "effq_fp_bucket_max": (_SZ, []),
    "effq_fp_bucket_ws_bytes": (_SZ, [_SZ]),
    "effq_fixed_point_bucket": (_I, [_P, _P, _P, _SZ, _I, _D, _D, _D, _I, _P, _P, _SZ, _P]),
    "effq_fp_bracket_ws_bytes": (_SZ, [_SZ]),
    "effq_fp_bracket_init": (_I, [_P, _P, _SZ, _I, _I, _P, _SZ, _P]),
    "effq_fp_bracket_run": (_I, [_P, _SZ, _I, _D, _D, _D, _I, _I, _P, _P, _P]),
    "effq_fp_bracket_stats": (_I, [_P, _SZ, _I, _D, _D, _P, _P, _P]),
    "effq_fp_bracket_update": (_I, [_SZ, _I, _D, _D, _D, _I, _P, _P, _P]),
    "effq_fp_traj_max": (_SZ, []),
    "effq_fp_traj_ws_bytes": (_SZ, [_SZ]),
    "effq_spd_inverse_prepare": (_I, [_P]),
    "effq_fp_traj_pred_bytes": (_SZ, []),
    "effq_fixed_point_traj": (_I, [_P, _P, _P, _SZ, _I, _D, _D, _D, _I, _P, _P, _P, _SZ, _P]),
    "effq_fixed_point_bucket_rec": (_I, [_P, _P, _P, _SZ, _I, _D, _D, _D, _I, _P, _P, _SZ, _P, _P]),
    "effq_fixed_point_coop_rec": (_I, [_P, _P, _P, _SZ, _I, _D, _D, _D, _I, _P, _P, _P, _P]),
    "effq_fp_check": (_I, [_P, _P, _P]),
    "effq_gram_packed_elems": (_SZ, [_I, _I]),
    "effq_gram_pack": (_I, [_P, _P, _I, _I, _P, _P]),
    "effq_gram_unpack": (_I, [_P, _I, _I, _P, _P, _P]),
    "effq_act_quant_backward": (_I, [_P, _P, _I, _P, _P, _P, _SZ, _P, _P]),
    "effq_att_classes_ws_bytes": (_SZ, []),
    "effq_att_classes": (_I, [_P, _LL, _P, _P, _P, C.POINTER(C.c_int32), _P, _P]),
    "effq_prof_enable": (_I, [_I]),
    "effq_prof_count": (_I, []),
    "effq_prof_read": (_I, [_I, C.POINTER(ProfRecord)]),
    "effq_admm_num_inverses": (_I, [_D, _D, _I, _I]),
    "effq_admm_run": (_I, [C.POINTER(AdmmRunArgs)]),
    "effq_admm_select_best": (_I, [_P, _I, _P, _P, _SZ, _SZ, _P, _P, _P, _P]),
    "effq_gram_ws_bytes": (_SZ, [_GP, _I]),
    "effq_gram_accum": (_I, [_P, _P, _P, _GP, _I, _P, _P, _I, _P, _SZ, _P]),
    "effq_gram_i8_supported": (_I, [_GP, _I]),
    "effq_gram_i8_ws_bytes": (_SZ, [_GP, _I]),
    "effq_gram_accum_i8": (_I, [_P, _P, _GP, _I, _P, _I, _P, _P, _P, _I, _LL, _P, _P, _I, _P, _SZ, _P]),
    "effq_gram_accum_i8_unw": (_I, [_P, _P, _GP, _I, _P, _I, _P, _P, _P, _I, _LL, _P, _P, _I, _P, _P, _P, _SZ, _P]),
    "effq_upsample_trilinear": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    "effq_gram_f64_supported": (_I, [_GP, _I]),
    "effq_gram_f64_ws_bytes": (_SZ, [_GP, _I]),
    "effq_gram_f64": (_I, [_P, _P, _GP, _I, _P, _P, _P, _SZ, _P]),
    "effq_gram_loss_ws_bytes": (_SZ, [_I]),
    "effq_gram_loss": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _SZ, _P]),
    "effq_packed_bytes": (_SZ, [_SZ, _I]),
    "effq_pack_levels": (_I, [_P, _SZ, _I, _P, _P]),
    "effq_unpack_levels": (_I, [_P, _SZ, _I, _P, _P]),
    "effq_ainv_ld": (_I, [_I]),
    "effq_fp_bracket_export_words": (_SZ, []),
    "effq_fp_bracket_export": (_I, [_P, _SZ, _P, _P, _SZ, _P]),
    "effq_fp_bracket_rebase": (_I, [_P, _P, _SZ, _P]),
    "effq_fp_bracket_import": (_I, [_P, _SZ, _P, _I, _SZ, _P, _P, _SZ, _P]),
    "effq_admm_uses_traj": (_I, [_SZ, _I]),
    "effq_gram_loss_i8_supported": (_I, [_I, _I, _I, _I]),
    "effq_gram_loss_i8_num_planes": (_I, [C.c_longlong]),
    "effq_gram_loss_i8_planes_bytes": (_SZ, [_I, _I, _I]),
    "effq_gram_loss_i8_prepare": (_I, [_P, _I, _I, _P, _I, _I, _P, _P, _P]),
    "effq_gram_loss_i8_ws_bytes": (_SZ, []),
    "effq_gram_loss_i8": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _SZ, _P]),
    "effq_spd_inverse_ws_bytes": (_SZ, [_I]),
    "effq_spd_inverse": (_I, [_P, _I, _I, _D, _D, _P, _P, _SZ, _P]),
    "effq_prox_ws_bytes": (_SZ, [_I, _I]),
    "effq_prox_solve": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _D, _D, _P, _P, _P, _SZ, _P]),
    "effq_prox_solve_shifted": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _D, _D, _D, _I, _P, _P, _P, _SZ, _P]),
    "effq_admm_presum": (_I, [_P, _P, _P, _SZ, _P]),
    "effq_admm_project_dual": (_I, [_P, _P, _P, _I, _P, _P, _F, _P, _SZ, _P]),
    "effq_conv_i8_supported": (_I, [_GP, _I, _I]),
    "effq_conv_i8s_supported": (_I, [_GP, _I, _I]),
    "effq_conv_i8s_ws_bytes": (_SZ, [_GP, _I, _I]),
    "conv3d_calib_step_i8s": (_I, [_P, _P, _P, _P, _GP, _P, _I, _P, _I, _I, _P, _P, _SZ, _P]),
    "effq_conv_i8_ws_bytes": (_SZ, [_GP]),
    "conv3d_calib_step_i8": (_I, [_P, _P, _P, _P, _GP, _P, _I, _P, _I, _P, _P, _SZ, _P]),
    "effq_conv_i8_out_supported": (_I, [_GP, _I, _I]),
    "conv3d_quant_forward_i8": (_I, [_P, _P, _P, _P, _P, _GP, _P, _I, _P, _I, _P, _P, _P, _SZ, _P]),
    "effq_conv_ws_bytes": (_SZ, [_GP]),
    "conv3d_quant_calib_step": (_I, [_P, _P, _P, _P, _P, _GP, _P, _I, _P, _P, _P, _SZ, _P]),
    "effq_adam_step": (_I, [_P, _P, _P, _P, _F, _F, _F, _F, _I, _SZ, _P]),
}

_lib = None


def load():
    """Load the shared library (once).  Raises EffqError when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EffqError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                        f"or `make -C efficientq_amd/csrc` (there is no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if a declared symbol is not exported
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != EFFQ_OK:
        msg = load().effq_last_error().decode(errors="replace")
        raise EffqError(f"{what} failed with {_ERR_NAMES.get(rc, rc)}: {msg}")
