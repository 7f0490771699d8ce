"""ctypes binding of libdwiseneuro_hip.so (C-ABI: include/dwn.h).

The product path has no fallback: if the shared library is missing or its ABI does not match, importing
this module raises.  Structures mirror include/dwn.h field by field; ``dwn_sizeof`` cross-checks the layout.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

# torch must be imported BEFORE the shared library is loaded: torch ships its own libamdhip64 and the process must
# hold exactly one HIP runtime.  With torch first, the library's DT_NEEDED libamdhip64.so.N binds to the copy torch
# already loaded (same SONAME); loaded the other way round the process ends up with two runtimes and every HIP call
# of this library fails with "no ROCm-capable device is detected".
import torch  # noqa: F401,E402

_HERE = Path(__file__).resolve().parent
# DWN_LIB_PATH: development override (A/B runs of two builds of the same ABI on one box)
# DWN_DETERMINISTIC=1: the ordered-reduction build of the same sources (csrc/Makefile, dwn_common.h) — run-to-run bit-identical
# results, several times slower; for re-run equality checks, not for training
DETERMINISTIC = os.environ.get("DWN_DETERMINISTIC", "0") == "1"
_LIB_NAME = "libdwiseneuro_hip_det.so" if DETERMINISTIC else "libdwiseneuro_hip.so"
LIB_PATH = Path(os.environ["DWN_LIB_PATH"]) if os.environ.get("DWN_LIB_PATH") else _HERE / "csrc" / _LIB_NAME

c_p = C.c_void_p
c_i = C.c_int
c_ll = C.c_longlong
c_f = C.c_float
c_d = C.c_double
c_sz = C.c_size_t

DWN_F32, DWN_BF16 = 0, 1
DWN_NREP = 32
LD_PLAIN, LD_PE, LD_BNACT, LD_AFFINE2, LD_DY3, LD_GATE, LD_CAT1 = 0, 1, 2, 3, 4, 5, 6
EPI_STORE, EPI_READOUT, EPI_DG, EPI_STORE_CAT, EPI_DH3 = 0, 1, 2, 3, 4
NN_AUTO, NN_XL128, NN_XL256, NN_TILE128, NN_KD = 0, 1, 2, 3, 4
F32_AUTO, F32_NATIVE, F32_SPLIT3 = 0, 1, 2
FAMILIES = ("pw_fwd", "dws_fwd", "dwt_fwd", "se_pool", "pwl_fwd", "resid_fwd", "resid_bwd", "pwl_dgrad", "pwl_wgrad",
            "bn3_reduce", "dwt_bwd", "dws_bwd", "pw_dgrad", "pw_wgrad", "cortex_fwd", "cortex_bwd", "readout_fwd",
            "readout_bwd")


class LoadDesc(C.Structure):
    _fields_ = [("p", c_p), ("q", c_p), ("ld", c_ll), ("v1", c_p), ("v2", c_p), ("v3", c_p), ("v4", c_p),
                ("v5", c_p), ("gate", c_p), ("gate2", c_p), ("gate_ld", c_i), ("rows_per_sample", c_i),
                ("act", c_i), ("pe_t", c_p), ("pe_h", c_p), ("pe_w", c_p), ("pT", c_i), ("pH", c_i),
                ("pW", c_i), ("pe_ld", c_i), ("ld2", c_ll), ("cat_c1", c_i), ("cat_c2", c_i)]


class GemmNNArgs(C.Structure):
    _fields_ = [("a", LoadDesc), ("a_kind", c_i), ("b", c_p), ("ldb", c_ll), ("c", c_p), ("ldc", c_ll),
                ("M", c_i), ("N", c_i), ("K", c_i), ("groups", c_i), ("stats", c_p),
                ("f32_split", c_i), ("stat_nchan", c_i), ("epi", c_i), ("bias", c_p),
                ("sp_beta", c_f), ("out_nct", c_p), ("Tn", c_i), ("n_valid", c_i), ("y3", c_p), ("ldy3", c_ll),
                ("s3", c_p), ("t3", c_p), ("dg", c_p), ("dg_ld", c_i), ("rows_per_sample", c_i),
                ("a2", c_p), ("a2_ld", c_ll), ("K1", c_i), ("b_sample_stride", c_ll), ("b_rows_per_sample", c_i),
                ("gate3", c_p), ("dps3", c_p), ("coef3", c_p), ("coef3_ld", c_i), ("variant", c_i)]


class GemmTNArgs(C.Structure):
    _fields_ = [("p", LoadDesc), ("p_kind", c_i), ("q", LoadDesc), ("q_kind", c_i), ("M", c_i), ("R", c_i),
                ("Cc", c_i), ("dw", c_p), ("lddw", c_ll), ("groups", c_i), ("rows_per_split", c_i),
                ("nsplit", c_i), ("R_load", c_i), ("rows_per_sample", c_i), ("splits_per_sample", c_i),
                ("dw_sample_stride", c_ll), ("overwrite", c_i), ("dw_f64", c_i)]


class DwSpatialFwdArgs(C.Structure):
    _fields_ = [("inp", LoadDesc), ("w", c_p), ("out", c_p), ("planes", c_i), ("Hin", c_i), ("Win", c_i),
                ("Hout", c_i), ("Wout", c_i), ("C", c_i), ("stride", c_i), ("ks", c_i), ("stats", c_p),
                ("rows_band", c_i), ("impl", c_i),
                ("a0", c_p), ("a0_ld", c_ll), ("w1", c_p), ("Cin", c_i)]      # rebuilt-input mode (include/dwn.h)


class DwSpatialBwdArgs(C.Structure):
    _fields_ = [("dy", LoadDesc), ("y1", LoadDesc), ("w", c_p), ("dh1", c_p), ("dw", c_p), ("planes", c_i),
                ("Hin", c_i), ("Win", c_i), ("Hout", c_i), ("Wout", c_i), ("C", c_i), ("stride", c_i),
                ("ks", c_i), ("stats", c_p), ("rows_band", c_i), ("impl", c_i),
                ("a0", c_p), ("a0_ld", c_ll), ("w1", c_p), ("Cin", c_i)]      # rebuilt-y1 mode (include/dwn.h)


class DwTemporalFwdArgs(C.Structure):
    _fields_ = [("inp", LoadDesc), ("w", c_p), ("out", c_p), ("B", c_i), ("T", c_i), ("HW", c_i), ("C", c_i),
                ("kt", c_i), ("stats", c_p), ("z_scale", c_p), ("z_shift", c_p), ("pooled", c_p)]


class DwTemporalBwdArgs(C.Structure):
    _fields_ = [("dy", LoadDesc), ("dy_kind", c_i), ("y2", LoadDesc), ("w", c_p), ("dh2", c_p), ("dw", c_p),
                ("B", c_i), ("T", c_i), ("HW", c_i), ("C", c_i), ("kt", c_i), ("stats", c_p)]


class BN(C.Structure):
    _fields_ = [("gamma", c_p), ("beta", c_p), ("running_mean", c_p), ("running_var", c_p),
                ("num_batches_tracked", c_p), ("coef", c_p), ("dgamma", c_p), ("dbeta", c_p)]


class StemArgs(C.Structure):
    _fields_ = [("dtype", c_i), ("training", c_i), ("B", c_i), ("Cin", c_i), ("C0", c_i), ("S", c_ll),
                ("eps", c_f), ("momentum", c_f), ("x", c_p), ("w", c_p), ("bn", BN),
                ("pe_t", c_p), ("pe_h", c_p), ("pe_w", c_p), ("T", c_i), ("H", c_i), ("W", c_i),
                ("y0", c_p), ("out", c_p),
                ("dout", c_p), ("dw", c_p), ("ws", c_p), ("ws_bytes", c_sz), ("xmom", c_p)]


class BlockArgs(C.Structure):
    _fields_ = [("dtype", c_i), ("training", c_i),
                ("B", c_i), ("T", c_i), ("Hin", c_i), ("Win", c_i), ("Hout", c_i), ("Wout", c_i), ("Cin", c_i),
                ("Cmid", c_i), ("Cout", c_i), ("stride", c_i), ("ks", c_i), ("kt", c_i), ("se_r", c_i),
                ("eps", c_f), ("momentum", c_f),
                ("x", c_p), ("out", c_p), ("y1", c_p), ("y2", c_p), ("y3", c_p), ("y4", c_p),
                ("z3", c_p), ("x_has_pe", c_i), ("a0", c_p),
                ("pe_t", c_p), ("pe_h", c_p), ("pe_w", c_p),
                ("out_pe_t", c_p), ("out_pe_h", c_p), ("out_pe_w", c_p),
                ("w_pw", c_p), ("w_dws", c_p), ("w_dwt", c_p), ("w_pwl", c_p), ("se_wr", c_p), ("se_br", c_p),
                ("se_we", c_p), ("se_be", c_p),
                ("bn1", BN), ("bn2", BN), ("bn3", BN), ("bn4", BN), ("bnsc", BN),
                ("drop_scale", c_p), ("hsrc", c_p), ("wsrc", c_p), ("hinv", c_p), ("winv", c_p),
                ("se_pmean", c_p), ("se_hidpre", c_p), ("se_gate", c_p),
                ("dout", c_p), ("dx", c_p), ("buf_a", c_p), ("buf_b", c_p), ("dy4", c_p), ("da0", c_p),
                ("dw_pw", c_p), ("dw_dws", c_p), ("dw_dwt", c_p), ("dw_pwl", c_p), ("dse_wr", c_p),
                ("dse_br", c_p), ("dse_we", c_p), ("dse_be", c_p),
                ("ws", c_p), ("ws_bytes", c_sz), ("pwl_bwd", c_i), ("f32_products", c_i), ("y1_mode", c_i)]


class PoolArgs(C.Structure):
    _fields_ = [("dtype", c_i), ("BT", c_ll), ("HW", c_i), ("C", c_i), ("x", c_p), ("out", c_p), ("dout", c_p),
                ("dx", c_p)]


class CortexArgs(C.Structure):
    _fields_ = [("dtype", c_i), ("training", c_i), ("B", c_i), ("T", c_i), ("Cin", c_i), ("C", c_i),
                ("groups", c_i), ("eps", c_f), ("momentum", c_f), ("x", c_p), ("out", c_p), ("y", c_p),
                ("w", c_p), ("bn", BN), ("bnsc", BN), ("drop_scale", c_p), ("dout", c_p), ("dx", c_p),
                ("dw", c_p), ("dout_mask", c_p), ("dout_mask_ld", c_i), ("ws", c_p), ("ws_bytes", c_sz), ("f32_products", c_i)]


class ReadoutArgs(C.Structure):
    _fields_ = [("dtype", c_i), ("B", c_i), ("T", c_i), ("Cin", c_i), ("groups", c_i), ("n_out", c_i),
                ("softplus_beta", c_f), ("x", c_p), ("w", c_p), ("bias", c_p), ("drop_mask", c_p), ("out", c_p),
                ("dout", c_p), ("dx", c_p), ("dw", c_p), ("dbias", c_p), ("ws", c_p), ("ws_bytes", c_sz), ("wt", c_p),
                ("f32_products", c_i)]


class TensorEntry(C.Structure):
    _fields_ = [("param", c_p), ("grad", c_p), ("exp_avg", c_p), ("exp_avg_sq", c_p), ("ema", c_p),
                ("numel", c_ll), ("is_int64", c_i), ("pad_", c_i)]


class ClipSrc(C.Structure):
    _fields_ = [("video", c_p), ("behavior", c_p), ("pupil_center", c_p), ("responses", c_p), ("length", c_ll),
                ("video_dtype", c_i), ("frame_start", c_i), ("frame_step", c_i), ("valid", c_i)]


class ClipDesc(C.Structure):
    _fields_ = [("src", ClipSrc), ("mix", ClipSrc), ("bbx1", c_i), ("bby1", c_i), ("bbx2", c_i), ("bby2", c_i),
                ("one_minus_lam", c_f), ("lam", c_f), ("mouse", c_i), ("mix_mode", c_i)]


class PwBwdArgs(C.Structure):
    _fields_ = [("dh1", c_p), ("a0", c_p), ("w_pw", c_p), ("abc", c_p), ("da0", c_p), ("dw", c_p),
                ("M", c_ll), ("E", c_i), ("Cin", c_i), ("ws", c_p), ("ws_bytes", c_sz), ("res", c_p), ("res_abc", c_p),
                ("res_C", c_i), ("res_hinv", c_p), ("res_winv", c_p), ("res_Hin", c_i), ("res_Win", c_i), ("res_Hout", c_i),
                ("res_Wout", c_i)]


class DwSpatialRcFwdArgs(C.Structure):
    _fields_ = [("a0", c_p), ("a0_ld", c_ll), ("blob", c_p), ("out", c_p), ("planes", c_i), ("Hin", c_i),
                ("Win", c_i), ("Hout", c_i), ("Wout", c_i), ("Cin", c_i), ("E", c_i), ("stride", c_i),
                ("stats", c_p), ("rows_band", c_i), ("round_y1", c_i)]


MIX_BOX, MIX_BLEND = 0, 1
VID_U8, VID_F32 = 0, 1

_STRUCTS = {
    "dwn_load_desc": LoadDesc, "dwn_gemm_nn_args": GemmNNArgs, "dwn_gemm_tn_args": GemmTNArgs,
    "dwn_dw_spatial_fwd_args": DwSpatialFwdArgs, "dwn_dw_spatial_bwd_args": DwSpatialBwdArgs,
    "dwn_dw_temporal_fwd_args": DwTemporalFwdArgs, "dwn_dw_temporal_bwd_args": DwTemporalBwdArgs,
    "dwn_bn": BN, "dwn_stem_args": StemArgs, "dwn_block_args": BlockArgs, "dwn_pool_args": PoolArgs,
    "dwn_cortex_args": CortexArgs, "dwn_readout_args": ReadoutArgs, "dwn_tensor_entry": TensorEntry,
    "dwn_clip_src": ClipSrc, "dwn_clip_desc": ClipDesc, "dwn_pw_bwd_args": PwBwdArgs,
    "dwn_dw_spatial_rc_fwd_args": DwSpatialRcFwdArgs,
}

# every symbol include/dwn.h declares: (restype, argtypes)
_P = C.POINTER
SYMBOLS = {
    "dwn_abi_version": (c_i, []),
    "dwn_source_hash": (C.c_char_p, []),
    "dwn_sizeof": (c_i, [C.c_char_p]),
    "dwn_last_error": (C.c_char_p, []),
    "dwn_profile_enable": (c_i, [C.c_ulonglong, c_i]),
    "dwn_profile_collect": (c_i, [c_i, _P(c_d), _P(c_ll)]),
    "dwn_gemm_nn": (c_i, [_P(GemmNNArgs), c_i, c_i, c_p]),
    "dwn_gemm_tn": (c_i, [_P(GemmTNArgs), c_i, c_i, c_p]),
    "dwn_dw_spatial_fwd": (c_i, [_P(DwSpatialFwdArgs), c_i, c_i, c_p]),
    "dwn_dw_spatial_bwd": (c_i, [_P(DwSpatialBwdArgs), c_i, c_i, c_p]),
    "dwn_dw_spatial_bwd_rc_supported": (c_i, [_P(DwSpatialBwdArgs), c_i]),
    "dwn_dw_spatial_fwd_rc_supported": (c_i, [_P(DwSpatialFwdArgs), c_i]),
    "dwn_dw_temporal_fwd": (c_i, [_P(DwTemporalFwdArgs), c_i, c_i, c_p]),
    "dwn_dw_temporal_bwd": (c_i, [_P(DwTemporalBwdArgs), c_i, c_i, c_p]),
    "dwn_bn_finalize": (c_i, [c_p, c_i, c_d, _P(BN), c_i, c_i, c_f, c_f, c_i, c_p]),
    "dwn_bn_bwd_finalize": (c_i, [c_p, c_d, _P(BN), c_p, c_i, c_i, c_p]),
    "dwn_pack_weight": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "dwn_stem_workspace_bytes": (c_sz, [_P(StemArgs)]),
    "dwn_stem_forward": (c_i, [_P(StemArgs), c_i, c_p]),
    "dwn_stem_backward": (c_i, [_P(StemArgs), c_i, c_p]),
    "dwn_block_workspace_bytes": (c_sz, [_P(BlockArgs), c_i]),
    "dwn_block_forward": (c_i, [_P(BlockArgs), c_i, c_p]),
    "dwn_block_backward": (c_i, [_P(BlockArgs), c_i, c_p]),
    "dwn_block_forward_writes": (c_i, [_P(BlockArgs)]),
    "dwn_pool_forward": (c_i, [_P(PoolArgs), c_i, c_p]),
    "dwn_pool_backward": (c_i, [_P(PoolArgs), c_i, c_p]),
    "dwn_cortex_workspace_bytes": (c_sz, [_P(CortexArgs), c_i]),
    "dwn_cortex_forward": (c_i, [_P(CortexArgs), c_i, c_p]),
    "dwn_cortex_backward": (c_i, [_P(CortexArgs), c_i, c_p]),
    "dwn_readout_workspace_bytes": (c_sz, [_P(ReadoutArgs), c_i]),
    "dwn_readout_wt_bytes": (c_sz, [_P(ReadoutArgs)]),
    "dwn_readout_forward": (c_i, [_P(ReadoutArgs), c_i, c_p]),
    "dwn_readout_backward": (c_i, [_P(ReadoutArgs), c_i, c_p]),
    "dwn_poisson_loss_forward": (c_i, [c_p, c_p, c_p, c_ll, c_ll, c_f, c_p, c_i, c_p]),
    "dwn_poisson_loss_backward": (c_i, [c_p, c_p, c_p, c_p, c_ll, c_ll, c_f, c_p, c_i, c_p]),
    "dwn_f64_to_f32": (c_i, [c_p, c_p, c_i, c_i, c_p]),
    "dwn_adamw_ema_multi": (c_i, [c_p, c_i, c_i, c_d, c_d, c_d, c_d, c_d, c_ll, c_d, c_d, c_i, c_p]),
    "dwn_ema_lerp_multi": (c_i, [c_p, c_i, c_i, c_d, c_i, c_p]),
    "dwn_conv_pw_bn_stats_workspace_bytes": (c_sz, [c_i]),
    "dwn_conv_pw_bn_stats": (c_i, [c_p, c_ll, c_ll, c_p, c_i, c_i, _P(BN), c_f, c_f, c_p, c_p, c_sz, c_i, c_i, c_p]),
    "dwn_pw_bwd_fused_supported": (c_i, [c_i, c_ll, c_i, c_i]),
    "dwn_pw_backward_workspace_bytes": (c_sz, [c_i, c_i, c_i]),
    "dwn_pw_backward": (c_i, [_P(PwBwdArgs), c_i, c_i, c_p]),
    "dwn_dw_spatial_rc_blob_bytes": (c_sz, [c_i, c_i]),
    "dwn_dw_spatial_rc_prep": (c_i, [c_p, c_p, c_p, c_i, c_i, c_p, c_i, c_p]),
    "dwn_dw_spatial_rc_supported": (c_i, [c_i, c_i, c_i, c_i, c_i, c_i, c_i]),
    "dwn_dw_spatial_fwd_rc": (c_i, [_P(DwSpatialRcFwdArgs), c_i, c_p]),
    "dwn_assemble_inputs": (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_p, c_i, c_p]),
    "dwn_assemble_targets": (c_i, [c_p, c_i, c_i, c_p, c_p, c_i, c_i, c_p, c_i, c_p]),
}


class DwnError(RuntimeError):
    pass


# csrc/Makefile HASH_SRCS, in its order
HASH_SRCS = ("dwn_api.hip", "dwn_gemm.hip", "dwn_gemm_xl.hip", "dwn_gemm_kd.hip", "dwn_dwconv.hip", "dwn_dwrc.hip", "dwn_dwbwd.hip", "dwn_dwfwd.hip",
             "dwn_elementwise.hip", "dwn_data.hip", "dwn_common.h", "dwn_internal.h", "dwn_kernels.h", "../../include/dwn.h")


def source_hash() -> str:
    import hashlib
    h = hashlib.sha256()
    for name in HASH_SRCS:
        h.update((_HERE / "csrc" / name).read_bytes())
    return h.hexdigest()[:16]


def _load():
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"(or `make -C {LIB_PATH.parent}`).  sensorium_amd has no fallback path.")
    lib = C.CDLL(str(LIB_PATH), mode=getattr(os, "RTLD_NOW", 2))
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    ab = bool(os.environ.get("DWN_LIB_PATH"))      # an explicitly chosen other build: a same-box A/B run of an older library
    if lib.dwn_abi_version() != 7 and not (ab and lib.dwn_abi_version() == 6):
        raise ImportError("libdwiseneuro_hip.so ABI version mismatch")
    built, have = lib.dwn_source_hash().decode(), source_hash()
    if built != have and not os.environ.get("DWN_LIB_PATH"):        # (an explicitly chosen other build is an A/B run)
        raise ImportError(f"{LIB_PATH.name} was built from other sources (binary {built}, tree {have}): rebuild it with "
                          f"`make -C {LIB_PATH.parent}` — binaries are not in git, so what runs must be what is committed")
    for cname, struct in _STRUCTS.items():
        n = lib.dwn_sizeof(cname.encode())
        if n != C.sizeof(struct) and not (ab and 0 < n < C.sizeof(struct)):      # (A/B: ABI 7 only appended a field)
            raise ImportError(f"struct layout mismatch for {cname}: C {n} bytes vs ctypes {C.sizeof(struct)}")
    return lib


lib = _load()


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib.dwn_last_error()
        raise DwnError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")
