"""ctypes binding of libmtdgan_hip.so (include/mtdgan_hip.h).  There is NO fallback: if the library is
missing or a call fails, this raises.  The CPU oracle under oracle/ is never imported from here."""
import ctypes as C
import os

import torch  # noqa: F401  -- must come first: torch's bundled HIP runtime has to be the one in the global scope

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmtdgan_hip.so")
def _lab_library():
    """The -DMTD_LAB build (_options.py, _build.py) for lab sessions: taken under MTD_LAB=1 when it exists, is not older than any
    kernel source (a stale lab build would silently measure yesterday's kernels) and MTD_LAB_LIB=0 does not ask for the shipped
    library with the Python-level lab switches only (bench.py's PMC child processes)."""
    lab = os.path.join(_HERE, "libmtdgan_hip_lab.so")
    if os.environ.get("MTD_LAB", "0") != "1" or os.environ.get("MTD_LAB_LIB", "1") == "0" or not os.path.exists(lab):
        return None
    csrc = os.path.join(_HERE, "csrc")
    newest = max(os.path.getmtime(os.path.join(csrc, f)) for f in os.listdir(csrc) if f.endswith((".hip", ".h")))
    return lab if os.path.getmtime(lab) >= newest else None


LIB_PATH = _lab_library() or LIB_PATH

ACT_NONE, ACT_RELU, ACT_LRELU, ACT_RELU_ADD = 0, 1, 2, 3


class Geom(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "B", "IH", "IW", "OH", "OW", "in_sy", "in_sx", "off_y", "off_x", "tap_dy", "tap_dx",
        "TH", "TW", "KW", "ky0", "kx0", "ky_step", "kx_step", "OHF", "OWF", "out_sy", "out_sx", "out_oy", "out_ox")]


class ConvArgs(C.Structure):
    _fields_ = [("g", Geom),
                ("inp", C.c_void_p), ("in_ld", C.c_int), ("C", C.c_int),
                ("w", C.c_void_p), ("w_sn", C.c_longlong), ("w_sc", C.c_longlong), ("w_st", C.c_longlong),
                ("N", C.c_int),
                ("out", C.c_void_p), ("out_ld", C.c_int),
                ("scale", C.c_void_p), ("bias", C.c_void_p),
                ("add1", C.c_void_p), ("add1_ld", C.c_int),
                ("add2", C.c_void_p), ("add2_ld", C.c_int),
                ("act", C.c_int),
                ("mask", C.c_void_p), ("mask_ld", C.c_int), ("mask_slope", C.c_float),
                ("ws", C.c_void_p), ("ws_bytes", C.c_size_t),
                ("scale2", C.c_void_p), ("scale_split", C.c_int),
                ("out2", C.c_void_p), ("out2_ld", C.c_int),
                ("tile_ctr", C.c_void_p), ("tile_ctr_len", C.c_int)]


class WgradArgs(C.Structure):
    _fields_ = [("g", Geom),
                ("p", C.c_void_p), ("p_ld", C.c_int), ("N", C.c_int),
                ("q", C.c_void_p), ("q_ld", C.c_int), ("C", C.c_int),
                ("dw", C.c_void_p), ("w_sn", C.c_longlong), ("w_sc", C.c_longlong),
                ("db", C.c_void_p), ("accumulate", C.c_int),
                ("ws", C.c_void_p), ("ws_bytes", C.c_size_t),
                ("half_scale", C.c_void_p), ("half_scale2", C.c_void_p), ("m_first", C.c_int)]


class WinoS2WeightDesc(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("sn", C.c_longlong), ("sc", C.c_longlong), ("st", C.c_longlong),
                ("N", C.c_int), ("C", C.c_int), ("groups", C.c_int), ("kmap", C.c_int * 16)]


class PackDesc(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("N", C.c_int), ("C", C.c_int), ("T", C.c_int),
                ("sn", C.c_longlong), ("sc", C.c_longlong)]


class WinoWeightDesc(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("sn", C.c_longlong), ("sc", C.c_longlong), ("st", C.c_longlong),
                ("N", C.c_int), ("C", C.c_int), ("kmap", C.c_int * 9), ("px", C.c_int)]


class SnLayer(C.Structure):
    _fields_ = [("w", C.c_void_p), ("u", C.c_void_p), ("v", C.c_void_p), ("sigma", C.c_void_p),
                ("u_save", C.c_void_p), ("v_save", C.c_void_p), ("rows", C.c_int), ("cols", C.c_int)]


class SnGradLayer(C.Structure):
    _fields_ = [("G", C.c_void_p), ("w", C.c_void_p), ("u", C.c_void_p), ("v", C.c_void_p), ("sigma", C.c_void_p),
                ("g_out", C.c_void_p), ("rows", C.c_int), ("cols", C.c_int), ("accumulate", C.c_int),
                ("G2", C.c_void_p), ("u2", C.c_void_p), ("v2", C.c_void_p), ("sigma2", C.c_void_p), ("prescaled", C.c_int),
                ("act_gy", C.c_void_p), ("act_gy2", C.c_void_p), ("act_a", C.c_void_p), ("act_bias", C.c_void_p),
                ("act_gy_ld", C.c_int), ("act_gy2_ld", C.c_int), ("act_a_ld", C.c_int), ("act_M", C.c_int), ("act_M_first", C.c_int),
                ("act_inv_slope", C.c_float)]


class WgradReduceDesc(C.Structure):
    _fields_ = [("a", WgradArgs), ("T", C.c_int), ("nslab", C.c_int), ("slab_stride", C.c_longlong),
                ("first_block", C.c_int), ("pad_", C.c_int)]


class MixReduceDesc(C.Structure):
    _fields_ = [("ws", C.c_void_p), ("dw2", C.c_void_p), ("db2", C.c_void_p), ("nslab", C.c_int), ("accumulate", C.c_int)]


class LossTerm(C.Structure):
    _fields_ = [("kind", C.c_int), ("a", C.c_void_p), ("b", C.c_void_p), ("tconst", C.c_float),
                ("mx", C.c_void_p), ("my", C.c_void_p), ("n", C.c_longlong), ("scale", C.c_float), ("eps", C.c_float),
                ("grad_out", C.c_void_p), ("coef", C.c_float), ("accumulate", C.c_int)]


class PtrPair(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p)]


class PatchDesc(C.Structure):
    _fields_ = [("slice", C.c_int), ("uy", C.c_float), ("ux", C.c_float), ("rot_k", C.c_int), ("flip", C.c_int), ("angle", C.c_float)]


class SumDesc(C.Structure):
    _fields_ = [("a", C.c_void_p), ("b", C.c_void_p), ("na", C.c_int), ("nb", C.c_int)]


class ZeroDesc(C.Structure):
    _fields_ = [("p", C.c_void_p), ("n", C.c_longlong)]


class AdamwTensor(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("n", C.c_longlong)]


_lib = None


def _preload_torch_hip_runtime():
    """libmtdgan_hip.so links libamdhip64.so.7 (ROCm), PyTorch ships its own libamdhip64.so.  Streams and
    device pointers come from PyTorch, so the kernels must be launched through PyTorch's runtime: load it
    into the global symbol scope before the library so every hip* symbol binds there."""
    tl = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(tl):
        C.CDLL(tl, mode=C.RTLD_GLOBAL)


RECORDER = None      # kernels.LaunchList: while a step is being recorded, the list every launch is appended to
_NOT_LAUNCHES = ("_half_scale_ok", "_pair_ok", "_group_ok", "_ws_bytes", "_blocks", "mtd_version", "_option", "mtd_lab_build", "mtd_prof_", "_override", "_bwd_ok", "_stamps", "_zmask_bytes", "_tail_ok", "_winograd_ok", "_winograd_s2_ok", "_weight_floats", "_kmap", "_plan_cfg", "_pair_ok", "_pair_mode", "_patch_w", "_f4_min_w", "_relu_add_ok")


class _RecordingLib:
    """Stand-in for the CDLL while kernels.LaunchList records a step: every launch entry point is called as usual and
    (function, arguments) is appended to the list.  Arguments are raw pointers, scalars and byref()s of argument structs
    (which keep their structs alive), so the pair can be called again as it is."""

    def __init__(self, L):
        self._L = L
        self._wrapped = {}

    def __getattr__(self, name):
        w = self._wrapped.get(name)
        if w is None:
            f = getattr(self._L, name)
            if any(t in name for t in _NOT_LAUNCHES):
                w = f
            else:
                def w(*args, _f=f):
                    rc = _f(*args)
                    if RECORDER is not None:
                        RECORDER.append((_f, args))
                    return rc
            self._wrapped[name] = w
        return w


_recording_lib = None


def lib():
    """Load the HIP library or fail loudly."""
    global _lib, _recording_lib
    if _lib is not None:
        if RECORDER is not None:
            if _recording_lib is None:
                _recording_lib = _RecordingLib(_lib)
            return _recording_lib
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the MTD-GAN HIP kernels are not built. Run `python -c 'import "
            "__graft_entry__ as g; g.build()'` (needs hipcc). There is no CPU fallback in this package.")
    _preload_torch_hip_runtime()
    L = C.CDLL(LIB_PATH)
    vp, ci, cf, ll, sz = C.c_void_p, C.c_int, C.c_float, C.c_longlong, C.c_size_t

    def sig(name, res, *args):
        f = getattr(L, name)
        f.restype = res
        f.argtypes = list(args)

    sig("mtd_version", C.c_char_p)
    sig("mtd_set_option", ci, C.c_char_p, ci)
    sig("mtd_get_option", ci, C.c_char_p, vp)
    sig("mtd_lab_build", ci)
    sig("mtd_conv_igemm_ws_bytes", sz, C.POINTER(ConvArgs))
    sig("mtd_conv_igemm", ci, C.POINTER(ConvArgs), vp)
    sig("mtd_conv_direct", ci, C.POINTER(ConvArgs), vp)
    sig("mtd_conv_relu_add_ok", ci, C.POINTER(ConvArgs))
    sig("mtd_conv_wgrad_ws_bytes", sz, C.POINTER(WgradArgs))
    sig("mtd_conv_wgrad", ci, C.POINTER(WgradArgs), vp)
    sig("mtd_conv_wgrad_half_scale_ok", ci, C.POINTER(WgradArgs))
    sig("mtd_conv_wgrad_plan_cfg", ci, C.POINTER(WgradArgs))
    sig("mtd_conv_wgrad_pair_ok", ci, C.POINTER(WgradArgs), ci)
    sig("mtd_conv_wgrad_pair_mode", ci, ci)
    sig("mtd_conv_wgrad_pair_ws_bytes", sz, C.POINTER(WgradArgs), ci)
    sig("mtd_conv_wgrad_pair", ci, C.POINTER(WgradArgs), vp, ci, vp)
    sig("mtd_conv_wgrad_pair_sum", ci, C.POINTER(WgradArgs), vp, vp, ci, vp)
    sig("mtd_conv_wgrad_slabs", ci, C.POINTER(WgradArgs), C.POINTER(C.c_int), C.POINTER(C.c_longlong), vp)
    sig("mtd_conv_wgrad_slabs_rfft", ci, C.POINTER(WgradArgs), C.POINTER(C.c_int), C.POINTER(C.c_longlong), vp, ci, vp, ci, ci, vp)
    sig("mtd_conv_wgrad_reduce_blocks", ci, C.POINTER(WgradReduceDesc))
    sig("mtd_conv_wgrad_reduce_multi", ci, vp, vp, ci, vp)
    sig("mtd_spec_mix_wgrad_reduce_multi", ci, vp, vp, ci, vp)
    sig("mtd_rfft_rows", ci, vp, ci, vp, ci, ci, vp)
    sig("mtd_spec_mix_fwd", ci, vp, vp, vp, vp, vp, vp, ci, vp)
    sig("mtd_spec_mix_bwd_ws_bytes", sz, ci)
    sig("mtd_spec_mix_bwd", ci, vp, vp, vp, vp, vp, vp, ci, vp)
    sig("mtd_spec_mix_wgrad_reduce", ci, vp, ci, vp, vp, ci, vp)
    sig("mtd_irfft_rows", ci, vp, vp, ci, vp, ci, vp, ci, vp, ci, ci, vp)
    sig("mtd_transpose64", ci, vp, vp, vp)
    sig("mtd_transpose64_multi", ci, vp, ci, vp)
    sig("mtd_rfft_rows_any", ci, vp, ci, vp, ci, ci, vp)
    sig("mtd_spec_mix_any", ci, vp, vp, vp, vp, ci, ci, vp)
    sig("mtd_irfft_rows_any", ci, vp, vp, ci, vp, ci, vp, ci, ci, ci, vp)
    sig("mtd_act_grad", ci, vp, ci, vp, ci, vp, ci, ll, ci, cf, vp)
    sig("mtd_copy_channels", ci, vp, ci, vp, ci, ll, ci, ci, vp)
    sig("mtd_upsample2x_fwd", ci, vp, ci, vp, ci, ci, ci, ci, ci, vp)
    sig("mtd_upsample2x_bwd", ci, vp, ci, vp, ci, ci, ci, ci, ci, vp)
    sig("mtd_upsample2x_bwd_masked", ci, vp, ci, vp, ci, vp, ci, cf, ci, ci, ci, ci, vp)
    sig("mtd_pixel_shuffle2_fwd", ci, vp, ci, vp, ci, ci, ci, ci, ci, vp)
    sig("mtd_pixel_shuffle2_bwd", ci, vp, ci, vp, ci, ci, ci, ci, ci, vp)
    sig("mtd_mul", ci, vp, vp, vp, ll, vp)
    sig("mtd_add", ci, vp, vp, vp, ll, vp)
    sig("mtd_dropout_mask", ci, vp, cf, cf, vp, ll, vp)
    sig("mtd_scale_by", ci, vp, vp, vp, ll, vp)
    sig("mtd_scalar_sums", ci, vp, ci, vp, vp)
    sig("mtd_zero_multi", ci, vp, vp, ci, vp)
    sig("mtd_checksum_multi", ci, vp, vp, ci, vp, vp)
    sig("mtd_pack_weights", ci, vp, vp, ci, vp)
    sig("mtd_upload", ci, vp, vp, sz, vp)
    sig("mtd_sn_ws_bytes", sz, vp, ci)
    sig("mtd_sn_power_iter", ci, vp, vp, ci, ci, vp, vp)
    sig("mtd_sn_power_iter_multi", ci, vp, vp, ci, ci, vp, vp)
    sig("mtd_sn_grad_ws_bytes", sz, vp, ci)
    sig("mtd_sn_grad", ci, vp, vp, ci, vp, vp)
    sig("mtd_pcgrad_ws_bytes", sz, ll, ci)
    sig("mtd_pcgrad_gram", ci, vp, vp, vp, vp, ci, ll, vp, vp, vp)
    sig("mtd_pcgrad_combine", ci, vp, vp, vp, vp, ci, ll, vp, vp, vp, vp, vp)
    sig("mtd_adamw_multi", ci, vp, vp, ci, cf, cf, cf, cf, cf, ci, vp)
    sig("mtd_adamw_multi_dyn", ci, vp, vp, ci, cf, cf, cf, vp, vp)
    sig("mtd_adamw_multi_pre", ci, vp, vp, ci, cf, cf, cf, cf, cf, cf, vp)
    sig("mtd_loss_terms_ws_bytes", sz, ci)
    sig("mtd_loss_terms", ci, vp, ci, vp, vp, vp)
    sig("mtd_loss_term_grads", ci, vp, ci, vp)
    sig("mtd_clip01", ci, vp, vp, ll, vp)
    sig("mtd_clip01_bwd", ci, vp, vp, vp, ll, vp)
    sig("mtd_edge_loss_ws_bytes", sz, ci)
    sig("mtd_image_metrics_ws_bytes", sz, ci, ci, ci)
    sig("mtd_image_metrics", ci, vp, vp, ci, ci, ci, ci, vp, vp, vp)
    sig("mtd_edge_loss", ci, vp, vp, ci, cf, cf, vp, vp, cf, ci, vp, vp)
    sig("mtd_foreground_bbox", ci, vp, ci, ci, ci, cf, vp, vp)
    sig("mtd_window_patches", ci, vp, vp, ci, ci, ci, vp, vp, ci, cf, cf, ci, vp, vp, vp)
    sig("mtd_hu_window", ci, vp, ll, cf, cf, vp, vp)
    sig("mtd_prof_mode", ci, ci)
    sig("mtd_spec_mix_zmask_bytes", sz, ci)
    sig("mtd_spec_mix_fwd4", ci, vp, vp, vp, vp, vp, vp, ci, vp)
    sig("mtd_spec_mix_bwd4", ci, vp, vp, vp, vp, vp, vp, ci, vp)
    sig("mtd_conv_c32_bwd_ok", ci, C.POINTER(ConvArgs), C.POINTER(WgradArgs))
    sig("mtd_conv_c32_bwd_ws_bytes", sz, C.POINTER(ConvArgs), C.POINTER(WgradArgs))
    sig("mtd_conv_c32_bwd", ci, C.POINTER(ConvArgs), C.POINTER(WgradArgs), C.POINTER(C.c_int), C.POINTER(C.c_longlong), vp)
    sig("mtd_conv_c32_bwd_irfft", ci, C.POINTER(ConvArgs), C.POINTER(WgradArgs), vp, C.POINTER(C.c_int), C.POINTER(C.c_longlong), vp)
    sig("mtd_conv_igemm_multi_ws_bytes", sz, C.POINTER(ConvArgs), ci)
    sig("mtd_conv_igemm_multi", ci, C.POINTER(ConvArgs), ci, vp)
    sig("mtd_resfft_block_tail_ok", ci, C.POINTER(ConvArgs))
    sig("mtd_resfft_block_tail", ci, C.POINTER(ConvArgs), vp, vp)
    sig("mtd_winograd_weight_floats", sz, ci, ci)
    sig("mtd_winograd_kmap", ci, C.POINTER(Geom), C.POINTER(C.c_int))
    sig("mtd_winograd_weights", ci, vp, vp, ci, vp)
    sig("mtd_conv_winograd_ok", ci, C.POINTER(ConvArgs))
    sig("mtd_conv_winograd_patch_w", ci, C.POINTER(ConvArgs))
    sig("mtd_conv_winograd_f4_min_w", ci, ci)
    sig("mtd_conv_winograd_ws_bytes", sz, C.POINTER(ConvArgs))
    sig("mtd_conv_winograd", ci, C.POINTER(ConvArgs), vp)
    sig("mtd_conv_winograd_group_ok", ci, C.POINTER(ConvArgs), ci)
    sig("mtd_conv_winograd_group", ci, C.POINTER(ConvArgs), ci, vp)
    sig("mtd_winograd_s2_kmap", ci, C.POINTER(Geom), C.POINTER(C.c_int), C.POINTER(C.c_int))
    sig("mtd_winograd_s2_weights", ci, vp, vp, ci, vp)
    sig("mtd_conv_winograd_s2_ok", ci, C.POINTER(ConvArgs), ci)
    sig("mtd_conv_winograd_s2_ws_bytes", sz, C.POINTER(ConvArgs), ci)
    sig("mtd_conv_winograd_s2", ci, C.POINTER(ConvArgs), ci, vp)
    sig("mtd_pcgrad_coeff", ci, vp, vp, ci, vp, vp)
    sig("mtd_pcgrad_axpy", ci, vp, vp, vp, vp, ci, ll, vp, cf, vp, vp)
    _lib = L
    return L


EXPORTS = [
    "mtd_version", "mtd_conv_igemm_ws_bytes", "mtd_conv_igemm", "mtd_conv_direct", "mtd_conv_wgrad_ws_bytes",
    "mtd_conv_wgrad", "mtd_conv_wgrad_half_scale_ok", "mtd_conv_wgrad_pair_ok", "mtd_conv_wgrad_pair_mode", "mtd_conv_wgrad_pair_ws_bytes", "mtd_conv_wgrad_pair", "mtd_conv_wgrad_pair_sum", "mtd_rfft_rows", "mtd_spec_mix_fwd", "mtd_spec_mix_bwd_ws_bytes", "mtd_spec_mix_bwd",
    "mtd_spec_mix_wgrad_reduce", "mtd_irfft_rows", "mtd_transpose64", "mtd_act_grad", "mtd_copy_channels",
    "mtd_upsample2x_fwd", "mtd_upsample2x_bwd", "mtd_pixel_shuffle2_fwd", "mtd_pixel_shuffle2_bwd", "mtd_mul", "mtd_pack_weights",
    "mtd_sn_ws_bytes", "mtd_sn_power_iter", "mtd_sn_power_iter_multi", "mtd_sn_grad_ws_bytes", "mtd_sn_grad", "mtd_pcgrad_ws_bytes",
    "mtd_pcgrad_gram", "mtd_pcgrad_combine", "mtd_adamw_multi", "mtd_adamw_multi_dyn", "mtd_adamw_multi_pre", "mtd_loss_terms_ws_bytes", "mtd_loss_terms",
    "mtd_loss_term_grads", "mtd_clip01", "mtd_clip01_bwd", "mtd_edge_loss_ws_bytes", "mtd_edge_loss",
    "mtd_prof_enable", "mtd_prof_collect", "mtd_conv_igemm_override", "mtd_conv_wgrad_override", "mtd_upload", "mtd_image_metrics_ws_bytes", "mtd_image_metrics", "mtd_rfft_rows_any", "mtd_spec_mix_any", "mtd_irfft_rows_any",
    "mtd_conv_wgrad_slabs", "mtd_conv_wgrad_slabs_rfft", "mtd_conv_wgrad_reduce_blocks", "mtd_conv_wgrad_reduce_multi", "mtd_spec_mix_wgrad_reduce_multi",
    "mtd_foreground_bbox", "mtd_window_patches", "mtd_hu_window", "mtd_add", "mtd_transpose64_multi", "mtd_upsample2x_bwd_masked",
    "mtd_prof_mode", "mtd_pcgrad_coeff", "mtd_pcgrad_axpy", "mtd_conv_igemm_multi_ws_bytes", "mtd_conv_igemm_multi",
    "mtd_conv_c32_bwd_ok", "mtd_conv_c32_bwd_ws_bytes", "mtd_conv_c32_bwd",
    "mtd_spec_mix_zmask_bytes", "mtd_spec_mix_fwd4", "mtd_spec_mix_bwd4",
    "mtd_resfft_block_tail_ok", "mtd_resfft_block_tail", "mtd_conv_c32_bwd_irfft",
    "mtd_winograd_weight_floats", "mtd_winograd_kmap", "mtd_winograd_weights", "mtd_conv_winograd_ok", "mtd_conv_winograd_ws_bytes", "mtd_conv_winograd", "mtd_conv_winograd_group_ok", "mtd_conv_winograd_group",
    "mtd_conv_wgrad_plan_cfg", "mtd_conv_winograd_patch_w", "mtd_conv_winograd_f4_min_w", "mtd_conv_relu_add_ok", "mtd_dropout_mask", "mtd_scale_by", "mtd_scalar_sums", "mtd_zero_multi", "mtd_checksum_multi",
    "mtd_set_option", "mtd_get_option", "mtd_lab_build",
    "mtd_winograd_s2_kmap", "mtd_winograd_s2_weights", "mtd_conv_winograd_s2_ok", "mtd_conv_winograd_s2_ws_bytes", "mtd_conv_winograd_s2",
]


class ProfRecord(C.Structure):
    _fields_ = [("kernel", C.c_int), ("cfg", C.c_int), ("splitk", C.c_int), ("N", C.c_int), ("C", C.c_int),
                ("taps", C.c_int), ("M", C.c_longlong), ("flops", C.c_double), ("ms", C.c_float), ("_pad", C.c_int),
                ("bytes", C.c_double)]


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"libmtdgan_hip: {what} failed with code {rc}")
