"""ctypes binding of libofq_hip.so (the C ABI declared in include/ofq_hip.h).

There is no CPU fallback: if the library is missing or a symbol is absent, importing the ops raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OFQ_HIP_LIB") or os.path.join(_HERE, "lib", "libofq_hip.so")   # override: kernel experiments only

i64, i32, f32, vp, sz = C.c_int64, C.c_int, C.c_float, C.c_void_p, C.c_size_t
f64 = C.c_double


class GemmDesc(C.Structure):
    _fields_ = [("A", vp), ("B", vp), ("C", vp), ("bias", vp),
                ("M", i64), ("N", i64), ("K", i64),
                ("lda", i64), ("ldb", i64), ("ldc", i64),
                ("transA", C.c_int32), ("transB", C.c_int32),
                ("nb0", C.c_int32), ("nb1", C.c_int32),
                ("sA0", i64), ("sA1", i64), ("sB0", i64), ("sB1", i64), ("sC0", i64), ("sC1", i64),
                ("nkb", C.c_int32), ("sAk", i64), ("sBk", i64),
                ("split_k", C.c_int32), ("alpha", f32), ("accumulate", C.c_int32), ("tile_hint", C.c_int32)]


class NtSeg(C.Structure):
    _fields_ = [("A", vp), ("B_bf16", vp), ("k_scale", vp), ("K", i64), ("lda", i64), ("ldb", i64), ("alpha", f32), ("amax", vp),
                ("hi_only", C.c_int32)]


class TnJob(C.Structure):
    _fields_ = [("dY", vp), ("codes", vp), ("dW", vp), ("lsq_s", vp), ("db", vp), ("baft", vp),
                ("S", i64), ("Ktok", i64), ("M", i64), ("N", i64), ("lda", i64), ("ldb", i64),
                ("gscale", f32), ("compute_db", C.c_int32), ("amax", vp)]


# name -> (restype, argtypes); must list EVERY symbol of include/ofq_hip.h (tests/test_abi.py checks)
SIGNATURES = {
    "ofq_abi_version": (i32, []),
    "ofq_source_hash": (C.c_char_p, []),
    "ofq_statsq_fwd": (i32, [vp, i64, i64, i32, vp, vp, vp, i32, i32, vp]),
    "ofq_statsq_codes_fwd": (i32, [vp, i64, i64, i32, vp, vp, vp, vp, vp, vp, vp]),
    "ofq_statsq_tensor_entry_bytes": (i64, []),
    "ofq_statsq_codes_multi": (i32, [vp, i64, vp]),
    "ofq_lsq_fwd": (i32, [vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, i32, i32, i32, f32, i32, vp]),
    "ofq_lsq_bwd_ws_bytes": (sz, [i64, i64, i64, i64, i32]),
    "ofq_lsq_fwd_patch": (i32, [vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, i32, i32, f32, i32, i32, i32, vp]),
    "ofq_lsq_bwd_patch": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, i32, i32, f32, i32, i32, i32, vp, sz, vp, vp]),
    "ofq_lsq_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, i32, i32, i32, f32, i32,
                          vp, sz, vp, vp]),
    "ofq_softmax_lsq_fwd": (i32, [vp, vp, vp, vp, i64, i64, i64, i64, f32, i32, f32, vp, vp, vp, i64, vp]),
    "ofq_softmax_lsq_bwd_ws_bytes": (sz, [i64]),
    "ofq_softmax_lsq_bwd": (i32, [vp, vp, vp, vp, vp, i64, i64, i64, i64, f32, i32, f32, vp, vp, sz, vp, vp]),
    "ofq_gemm_ws_bytes": (sz, [C.POINTER(GemmDesc)]),
    "ofq_gemm_f32": (i32, [C.POINTER(GemmDesc), vp, sz, vp]),
    "ofq_qgemm_i8_nt": (i32, [vp, vp, vp, vp, vp, f32, vp, vp, i64, f32, i64, i64, i64, i64, i64, i64, vp]),
    "ofq_qgemm_i8_nt_q": (i32, [vp, vp, vp, vp, vp, f32, vp, vp, i64, f32, i64, i64, i64, i64, i64, i64,
                                vp, i64, vp, i64, f32, vp, i32, i32, i32, i32, i64, i32, vp]),
    "ofq_qgemm_i8_lsq_bwd_ws_bytes": (sz, [i64, i64, i32]),
    "ofq_qgemm_i8_lsq_bwd": (i32, [vp, vp, vp, vp, f32, vp, vp, i64, f32, i64, i64, i64, i64, i64, vp, i64, vp, i64,
                                   vp, i64, f32, vp, i32, i32, i32, i32, i64, i32, vp, vp, vp, vp, sz, vp, vp]),
    "ofq_attn_f32_fwd": (i32, [vp, vp, i64, i64, i64, i64, f32, vp]),
    "ofq_qattn_dqkx_lsq_bwd": (i32, [vp, vp, vp, vp, f32, vp, vp, f32, vp, vp, i64, vp, i64, i64, i64, i64, i64, i64, vp, i64,
                                     vp, i64, f32, vp, i32, i32, vp, vp, vp, vp, sz, vp, vp]),
    "ofq_qgemm_bf16s_nt": (i32, [vp, vp, vp, vp, f32, i32, i32, i64, i64, i64, i64, i64, i64, vp, vp, vp, vp]),
    "ofq_qgemm_bf16s_nt_sk_ws_bytes": (sz, [i32]),
    "ofq_qgemm_bf16s_nt_sk_pays": (i32, [i64, i64, i64, i32]),
    "ofq_qgemm_bf16s_nt_sk": (i32, [C.POINTER(NtSeg), i32, vp, i32, i64, i64, i64, i32, vp, sz, vp, vp]),
    "ofq_qgemm_bf16s_nt_sk_check": (i32, [vp, vp, vp]),
    "ofq_qgemm_bf16s_nt_sk_reset": (i32, [vp, vp]),
    "ofq_qgemm_bf16s_nt_lsq_ws_bytes": (sz, [i64, i64]),
    "ofq_qgemm_bf16s_nt_lsq": (i32, [vp, vp, vp, f32, vp, vp, i64, f32, vp, i32, i32, i32, vp, vp, vp, vp, i64, i64, i64, i64,
                                     i64, i64, vp, sz, vp, vp]),
    "ofq_qgemm_bf16s_tn_ws_bytes": (sz, [i64, i64, i32]),
    "ofq_qgemm_bf16s_tn": (i32, [vp, vp, vp, vp, i64, f32, vp, i32, vp, i64, i64, i64, i64, i64, i32, vp, sz, vp, vp]),
    "ofq_qgemm_bf16s_tn_group_ws_bytes": (sz, [C.POINTER(TnJob), i32, i32]),
    "ofq_qgemm_bf16s_tn_group": (i32, [C.POINTER(TnJob), i32, i32, vp, sz, vp]),
    "ofq_codes_transpose_bf16": (i32, [vp, vp, i64, i64, vp]),
    "ofq_codes_transpose_f16": (i32, [vp, vp, i64, i64, vp]),
    "ofq_absmax_f32": (i32, [vp, i64, i64, i64, vp, vp]),
    "ofq_lsq_eff_scale_vec": (i32, [vp, f32, vp, i64, i64, vp]),
    "ofq_rowdot_i8": (i32, [vp, vp, vp, i64, i64, vp]),
    "ofq_qattn_scores_i8": (i32, [vp, vp, vp, vp, f32, vp, f32, vp, vp, vp, i64, i64, i64, i64, i64, vp]),
    "ofq_qattn_scores_plain_i8": (i32, [vp, vp, vp, vp, f32, vp, f32, vp, vp, vp, i64, i64, i64, i64, i64, vp]),
    "ofq_qattn_scores_softmax_i8": (i32, [vp, vp, vp, f32, vp, f32, vp, vp, vp, i32, vp, f32, f32, i32, vp, i64, vp, vp, vp,
                                          i64, i64, i64, i64, i64, vp]),
    "ofq_qattn_dq_plain_bf16s": (i32, [vp, vp, vp, vp, f32, i64, i64, i64, i64, i64, vp]),
    "ofq_qattn_dk_plain_bf16s": (i32, [vp, vp, vp, vp, f32, vp, i64, i64, i64, i64, i64, vp]),
    "ofq_qattn_pv_i8": (i32, [vp, vp, vp, vp, f32, vp, f32, vp, vp, i64, i64, i64, i64, i64, vp]),
    "ofq_qattn_dp_bf16s": (i32, [vp, vp, vp, vp, f32, vp, i64, i64, i64, i64, i64, vp]),
    "ofq_qattn_dv_bf16s": (i32, [vp, vp, vp, vp, f32, i64, i64, i64, i64, i64, vp]),
    "ofq_qattn_dp_softmax_bwd_ws_bytes": (sz, [i64, i64, i64]),
    "ofq_qattn_dp_softmax_bwd": (i32, [vp, vp, vp, f32, vp, vp, vp, f32, f32, i32, vp, vp, vp, i64, i64, i64, i64, i64, vp, sz, vp, vp]),
    "ofq_qattn_dqkx_bf16s": (i32, [vp, vp, vp, vp, f32, vp, i64, i64, i64, i64, i64, vp, vp]),
    "ofq_qattn_dxq_bf16s": (i32, [vp, vp, vp, vp, f32, i32, i64, i64, i64, i64, i64, vp, vp]),
    "ofq_rowdot_i8_multi": (i32, [vp, vp, vp, i64, i64, i32, vp]),
    "ofq_rowdot_f32_seg": (i32, [vp, vp, vp, i64, i32, i32, i64, vp]),
    "ofq_codes_transpose_i8": (i32, [vp, vp, i64, i64, i64, i64, vp]),
    "ofq_qattn_prep": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, vp]),
    "ofq_colsum_ws_bytes": (sz, [i64, i64]),
    "ofq_colsum": (i32, [vp, vp, i64, i64, i64, vp, sz, vp]),
    "ofq_layernorm_fwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, f32, vp, vp]),
    "ofq_layernorm_bwd_ws_bytes": (sz, [i64, i64]),
    "ofq_layernorm_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, vp, sz, vp, vp]),
    "ofq_layernorm_lsq_fwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, f32, vp, i32, i32, i64, i64, i64, f32, vp]),
    "ofq_layernorm_lsq_bwd_ws_bytes": (sz, [i64, i64]),
    "ofq_layernorm_lsq_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i64, f32, vp, i32, i32, vp, vp, vp, vp, vp, vp, i64, i64,
                                    i64, i64, vp, sz, vp, vp]),
    "ofq_layernorm_lsq_fwd_perm": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, f32, vp, i32, i32, i64, i64, i64, f32,
                                         vp, vp, i64, vp]),
    "ofq_layernorm_lsq_bwd_perm": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i64, f32, vp, i32, i32, vp, vp, vp, vp, vp, vp, i64, i64,
                                         i64, i64, vp, sz, vp, vp, vp, i64, vp, vp]),
    "ofq_adamw_tensor_entry_bytes": (i64, []),
    "ofq_adamw_multi": (i32, [vp, i64, f32, f64, f64, f32, f32, f64, f64, vp]),
    "ofq_adamw_hyper_pack": (i32, [vp, f32, f64, f64, f32, f32, f64, f64]),
    "ofq_adamw_multi_dev": (i32, [vp, i64, vp, vp]),
    "ofq_adamw_multi_g": (i32, [vp, i64, f32, f64, f64, f32, f32, f64, f64, vp, vp]),
    "ofq_adamw_multi_dev_g": (i32, [vp, i64, vp, vp, vp]),
    "ofq_step_guard": (i32, [vp, i32, vp, vp, vp, vp]),
    "ofq_store_f32": (i32, [vp, vp, i32, vp]),
    "ofq_cga_freeze_mask": (i32, [vp, i64, i64, i32, f32, vp, vp, vp]),
    "ofq_cga_tensor_entry_bytes": (i64, []),
    "ofq_cga_freeze_mask_multi": (i32, [vp, i64, i32, f32, vp]),
    "ofq_cga_mask_grad_save": (i32, [vp, vp, vp, vp, i64, vp]),
    "ofq_cga_restore": (i32, [vp, vp, vp, i64, vp]),
    "ofq_sum_defer": (None, [i32]),
    "ofq_sum_pending": (i32, []),
    "ofq_sum_flush": (i32, [vp]),
    "ofq_gelu_fwd": (i32, [vp, vp, i64, vp, vp]),
    "ofq_permute_tokens": (i32, [vp, vp, vp, i64, i64, i64, vp]),
    "ofq_kd_loss_fwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, vp]),
    "ofq_kd_loss_bwd": (i32, [vp, vp, vp, vp, vp, vp, i64, vp]),
    "ofq_assemble_tokens": (i32, [vp, vp, vp, vp, vp, i64, i64, i64, vp]),
    "ofq_split_f32_bf16x3": (i32, [vp, vp, i64, i64, vp]),
    "ofq_gemm_bf16x3x3_nt": (i32, [vp, vp, vp, vp, i32, i64, i64, i64, i64, i64, i64, i64, vp]),
    "ofq_input_pipeline_u8": (i32, [vp, vp, i64, i64, i64, i64, vp, vp, i32, i32, f32, f32, i32, i32, i32, i32, vp, vp, vp]),
}

_lib = None
ABI_VERSION = 2


def load():
    """Load the shared library; it is (re)built first when it is missing or was built from other sources than the ones
    next to it (content hash, ofq_amd/build.py), under a file lock so that ranks started together do not race."""
    global _lib
    if _lib is not None:
        return _lib
    if "OFQ_HIP_LIB" not in os.environ:
        from . import build as _b
        if _b.needs_build():
            try:
                _b.build()
            except Exception as e:  # noqa: BLE001
                raise RuntimeError("ofq_amd: %s is missing or stale (built from %s, sources are %s) and could not be "
                                   "rebuilt (%s). Run `python -m ofq_amd.build`; there is no CPU fallback."
                                   % (LIB_PATH, _b.built_hash(), _b.source_hash(), e))
    # PyTorch-ROCm ships its own libamdhip64; it has to be in the process BEFORE this library is mapped, otherwise the
    # loader binds our kernels to /opt/rocm's copy and every launch fails with hipErrorNoDevice (two HIP runtimes, the
    # streams and allocations belong to torch's).  Importing torch first makes the soname resolve to the loaded one.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    if lib.ofq_abi_version() != ABI_VERSION:
        raise RuntimeError("ofq_amd: ABI version mismatch in %s" % LIB_PATH)
    _lib = lib
    return lib
