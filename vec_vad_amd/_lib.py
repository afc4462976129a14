"""ctypes binding of libvecvad_hip.so (the C ABI declared in include/vecvad_hip.h).

There is deliberately NO fallback: if the library is missing or a call returns a non-zero status this module
raises.  The reference's equivalent seam is the cffi glue of the FlowNet2 ops
(FlowNet2_src/models/components/ops/*/functions/*.py) -- python allocates tensors, C launches on the current stream.
"""
import ctypes as C
import os

import torch  # noqa: F401  -- MUST be imported before the .so: both link libamdhip64; loading PyTorch's copy first makes
#                       the dynamic linker bind our kernels to the same HIP runtime (one device context, shared streams)

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('VV_LIB_PATH') or os.path.join(HERE, 'csrc', 'libvecvad_hip.so')      # VV_LIB_PATH: A/B experiments (tools/)

c_i32, c_i64, c_f32, c_f64, c_vp = C.c_int32, C.c_int64, C.c_float, C.c_double, C.c_void_p

IN_PLAIN, IN_ACT, IN_POOL, IN_CAT, IN_CUBE = 0, 1, 2, 3, 4
CONV3, CONVT_FWD, CONVT_DGRAD = 0, 1, 2
CONV_BF16 = 1          # vv_conv_params.pad0 flag (include/vecvad_hip.h VV_CONV_BF16)
CONV_SRC_BF16 = 2      # VV_CONV_SRC_BF16
PACK_BF16 = 4          # vv_pack_entry.mode flag: bf16 panel for VV_CONV_BF16 launches
BNBWD_DZ_BF16 = 1      # vv_bnbwd_params.flags
BNBWD_PARTIALS_PER_CUBE = 2
BNBWD_DA_BF16 = 4
CONV_OUT_BF16 = 8       # VV_CONV_OUT_BF16
CONV_ALLSRC_BF16 = 16   # VV_CONV_ALLSRC_BF16
CONV_RELU = 32          # VV_CONV_RELU: eval mode, BatchNorm folded into the filter, ReLU in the epilogue
CONV_NO_GEMM16 = 64     # VV_CONV_NO_GEMM16: keep an all-bf16 3x3 launch on the round-3 kernel (A/B switch)
CONV_NO_RING = 128      # VV_CONV_NO_RING: keep the 32x32-level Winograd launches on the per-tile kernel (A/B switch)
BNBWD_Y_BF16 = 8
BNBWD_PARTIALS_PER_TILE = 16   # partial sums left by the data-gradient launch (ConvParams.bn_partial)
BNBWD_PARTIALS_PER_CTILE = 32  # ... of vv_conv_mfma (all-bf16 tensors): rows per vv_conv_ntiles
BNBWD_PARTIALS_PER_TILE44 = 64  # ... of vv_conv_wino44: rows per vv_wino44_ntiles
BNBWD_PARTIALS_PER_TTILE = 128  # ... of the transposed conv's data gradient (fp32 kernel): rows per vv_convt_dgrad_ntiles(B, H, W, 0)
WGRAD_X_BF16 = 2
WGRAD_DY_BF16 = 1      # vv_wgrad_params.pad0 for vv_wgrad_bf16


class View(C.Structure):
    _fields_ = [('ptr', c_vp), ('gstride', c_i64), ('cstride', c_i32), ('coff', c_i32)]


class ConvParams(C.Structure):
    _fields_ = [('kind', c_i32), ('in_mode', c_i32), ('G', c_i32), ('B', c_i32), ('H', c_i32), ('W', c_i32),
                ('Cin', c_i32), ('CinP', c_i32), ('Cout', c_i32),
                ('src0', View), ('a', c_vp), ('b', c_vp), ('ab_gstride', c_i64),
                ('src1', View), ('csplit', c_i32), ('pad0', c_i32), ('chmap', c_vp),
                ('w', c_vp), ('w_gstride', c_i64), ('bias', c_vp), ('bias_gstride', c_i64),
                ('out', View), ('stats', c_vp),
                ('bn_z', c_vp), ('bn_z_gstride', c_i64), ('bn_a', c_vp), ('bn_b', c_vp), ('bn_mean', c_vp), ('bn_invstd', c_vp),
                ('bn_gstride', c_i64), ('bn_partial', c_vp),
                ('out1', View), ('osplit', c_i32), ('pad1', c_i32)]


class WgradParams(C.Structure):
    _fields_ = [('kind', c_i32), ('in_mode', c_i32), ('G', c_i32), ('B', c_i32), ('H', c_i32), ('W', c_i32),
                ('Cin', c_i32), ('CinP', c_i32), ('Cout', c_i32), ('ksplit', c_i32),
                ('src0', View), ('a', c_vp), ('b', c_vp), ('ab_gstride', c_i64),
                ('src1', View), ('csplit', c_i32), ('pad0', c_i32), ('chmap', c_vp),
                ('dy', View), ('partial', c_vp), ('partial_gstride', c_i64)]


class PackEntry(C.Structure):
    _fields_ = [('src_off', c_i64), ('dst_off', c_i64), ('mode', c_i32), ('K', c_i32), ('KP', c_i32), ('N', c_i32)]


class ReduceEntry(C.Structure):
    _fields_ = [('kind', c_i32), ('Cin', c_i32), ('Cout', c_i32), ('NCO', c_i32), ('nslab', c_i32), ('block_start', c_i32),
                ('part_off', c_i64), ('grad_off', c_i64), ('grad_gstride', c_i64)]


class FoldEntry(C.Structure):
    _fields_ = [('w_off', c_i64), ('b_off', c_i64), ('g_off', c_i64), ('beta_off', c_i64), ('rm_off', c_i64), ('rv_off', c_i64),
                ('cout', c_i32), ('row', c_i32)]


class BnBwdParams(C.Structure):
    _fields_ = [('G', c_i32), ('B', c_i32), ('H', c_i32), ('W', c_i32), ('C', c_i32), ('flags', c_i32),
                ('y', c_vp), ('y_gstride', c_i64),
                ('a', c_vp), ('b', c_vp), ('mean', c_vp), ('invstd', c_vp), ('ab_gstride', c_i64),
                ('dA', View), ('dpool', c_vp), ('dpool_gstride', c_i64),
                ('dz', c_vp), ('dz_gstride', c_i64), ('partial', c_vp)]


class OutconvParams(C.Structure):
    _fields_ = [('G', c_i32), ('B', c_i32), ('HW', c_i32), ('C', c_i32),
                ('y', c_vp), ('y_gstride', c_i64), ('a', c_vp), ('b', c_vp), ('ab_gstride', c_i64),
                ('w', c_vp), ('bias', c_vp), ('param_gstride', c_i64), ('oc', c_vp),
                ('tgt0', c_vp), ('tgt0_cstride', c_i32), ('pad0', c_i32),
                ('tgt1', c_vp), ('tgt1_cstride', c_i32), ('pad1', c_i32),
                ('tgt_src', c_vp), ('tgt_coff', c_vp), ('out4', c_vp), ('score', c_vp), ('gscale', c_vp),
                ('dout4', c_vp)]


class Conv2dParams(C.Structure):
    _fields_ = [('kind', c_i32), ('R', c_i32), ('stride', c_i32), ('B', c_i32), ('H', c_i32), ('W', c_i32),
                ('Cin', c_i32), ('CinP', c_i32), ('Cout', c_i32), ('CoutP', c_i32), ('src', View), ('w', c_vp),
                ('bias', c_vp), ('slope', c_f32), ('pad0', c_i32), ('out', View)]


_SIGS = {
    'vv_conv2d_mfma': (c_i32, [C.POINTER(Conv2dParams), c_vp]),
    'vv_conv2d_wino': (c_i32, [c_vp, c_i32, c_i32, c_i64, c_vp, c_vp, c_f32, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    'vv_conv2d_splitk_finish': (c_i32, [c_vp, c_i32, c_i64, c_i32, c_i32, c_vp, c_f32, c_vp, c_i32, c_i32, c_vp]),
    'vv_pack_conv2d': (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    'vv_conv3x3_n2': (c_i32, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_i32, c_vp, c_f32, c_vp, c_i32, c_i32, c_vp]),
    'vv_deconv4x4_c2': (c_i32, [c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_f32, c_vp, c_i32, c_i32, c_vp]),
    'vv_upsample4': (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_f32, c_vp]),
    'vv_conv_mfma': (c_i32, [C.POINTER(ConvParams), c_vp]),
    'vv_conv_wino': (c_i32, [C.POINTER(ConvParams), c_vp]),
    'vv_wino_ntiles': (c_i32, [c_i32, c_i32]),
    'vv_pack_wino': (c_i32, [c_vp, c_i32, c_i32, c_vp, c_i64, c_vp, c_i64, c_i32, c_vp]),
    'vv_conv_wino44': (c_i32, [C.POINTER(ConvParams), c_vp]),
    'vv_wino44_ntiles': (c_i32, [c_i32, c_i32]),
    'vv_pack_wino44': (c_i32, [c_vp, c_i32, c_i32, c_vp, c_i64, c_vp, c_i64, c_i32, c_vp]),
    'vv_conv_ntiles': (c_i32, [c_i32, c_i32, c_i32]),
    'vv_conv_ntiles2': (c_i32, [c_i32, c_i32, c_i32, c_i32, c_i32]),
    'vv_convt_dgrad_ntiles': (c_i32, [c_i32, c_i32, c_i32, c_i32]),
    'vv_wgrad_mfma': (c_i32, [C.POINTER(WgradParams), c_vp]),
    'vv_wgrad_ntiles': (c_i32, [c_i32, c_i32, c_i32, c_i32]),
    'vv_wgrad_bf16': (c_i32, [C.POINTER(WgradParams), c_vp]),
    'vv_wgrad_bf16_plan': (c_i32, [c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, C.POINTER(c_i32), C.POINTER(c_i32), C.POINTER(c_i32)]),
    'vv_wgrad_reduce': (c_i32, [c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_i64, c_vp, c_i64, c_vp]),
    'vv_wgrad_reduce_grouped': (c_i32, [c_vp, c_i32, c_i32, c_i32, c_vp, c_i64, c_vp, c_vp]),
    'vv_pack_weights': (c_i32, [c_vp, c_i32, c_i32, c_vp, c_i64, c_vp, c_i64, c_i32, c_vp]),
    'vv_bn_finalize': (c_i32, [c_i32, c_i32, c_i32, c_i64, c_i32, c_f32, c_f32, c_vp, c_i64, c_vp, c_vp, c_i64,
                               c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    'vv_bn_bwd_reduce': (c_i32, [C.POINTER(BnBwdParams), c_vp]),
    'vv_bn_bwd_nblk': (c_i32, [c_i32, c_i32, c_i32, c_i32]),
    'vv_bn_bwd_apply': (c_i32, [C.POINTER(BnBwdParams), c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_vp]),
    'vv_outconv_fwd': (c_i32, [C.POINTER(OutconvParams), c_vp]),
    'vv_outconv_bwd': (c_i32, [c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64,
                               c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp]),
    'vv_outconv_bwd_nblk': (c_i32, [c_i32, c_i32]),
    'vv_outconv_fwdbwd': (c_i32, [C.POINTER(OutconvParams), c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp]),
    'vv_outconv_bwd_reduce': (c_i32, [c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    'vv_bias_grad': (c_i32, [c_i32, c_i64, c_i32, c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_i64, c_vp]),
    'vv_bias_from_partials': (c_i32, [c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_i64, c_vp, c_i64, c_vp]),
    'vv_adam': (c_i32, [c_i64, c_vp, c_vp, c_vp, c_vp, c_f32, c_f32, c_f32, c_f32, c_f32, c_f32, c_f32, c_vp]),
    'vv_adam_tick': (c_i32, [c_vp, c_f32, c_f64, c_f64, c_vp, c_vp]),
    'vv_counter_add': (c_i32, [c_vp, c_i32, c_i64, c_vp]),
    'vv_adam_bucketed': (c_i32, [c_i32, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_f32, c_f32, c_f32, c_f32, c_vp]),
    'vv_cube_gather': (c_i32, [c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'vv_pool_act': (c_i32, [c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64, c_i32, c_vp]),
    'vv_cube_erase': (c_i32, [c_i32, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp, c_i64, c_i32, c_vp]),
    'vv_nchw_to_nhwc': (c_i32, [c_i32, c_i32, c_i32, c_vp, c_vp, c_vp]),
    'vv_out4_to_nchw': (c_i32, [c_i32, c_i32, c_i32, c_vp, c_vp, c_i32, c_i32, c_vp]),
    'vv_nchw_to_out4': (c_i32, [c_i32, c_i32, c_i32, c_vp, c_i32, c_i32, c_vp, c_vp]),
    'vv_correlation_fwd': (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32,
                                   c_i32, c_vp]),
    'vv_correlation_out_shape': (c_i32, [c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32,
                                         C.POINTER(c_i32), C.POINTER(c_i32), C.POINTER(c_i32)]),
    'vv_correlation_nhwc': (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_i32, c_i32, c_f32, c_vp]),
    'vv_resample2d_fwd': (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    'vv_channelnorm_fwd': (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    'vv_fold_bn': (c_i32, [c_vp, c_i32, c_i32, c_vp, c_i64, c_vp, c_i64, c_f32, c_vp, c_i64, c_vp]),
    'vv_flownet_prep_workspace_bytes': (C.c_int64, [c_i32]),
    'vv_flownet_prep': (c_i32, [c_vp, c_i32, c_i32, c_i32, c_f32, c_vp, C.c_int64, c_vp, c_vp, c_vp, c_vp]),
    'vv_warp_pack12': (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32, c_f32, c_vp, c_vp]),
    'vv_fusion_pack11': (c_i32, [c_vp, c_vp, c_vp, c_i32, c_vp, c_i32, c_i32, c_i32, c_i32, c_f32, c_vp, c_vp]),
    'vv_crop_resize': (c_i32, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp]),
    'vv_frame_scores': (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, C.c_double, C.c_double, C.c_double, c_i32, c_vp,
                                c_vp]),
    'vv_roc_auc_counts': (c_i32, [c_vp, c_vp, c_i32, c_vp, c_vp]),
    'vv_version': (C.c_char_p, []),
    'vv_abi_sizeof': (c_i32, [c_i32]),
    'vv_status_string': (C.c_char_p, [c_i32]),
    'vv_device_arch_ok': (c_i32, []),
    'vv_num_cus': (c_i32, []),
}

EXPORTS = sorted(_SIGS)
# order = the `which` argument of vv_abi_sizeof
ABI_STRUCTS = (View, ConvParams, WgradParams, PackEntry, ReduceEntry, FoldEntry, BnBwdParams, OutconvParams, Conv2dParams)
_lib = None


class VecVadHipError(RuntimeError):
    pass


def lib():
    """Load the shared library (once).  Raises if it has not been built -- there is no CPU/PyTorch fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VecVadHipError('libvecvad_hip.so is missing (%s). Build it with `python -m vec_vad_amd.build` '
                                 '(hipcc --offload-arch=gfx950); vec_vad_amd has no fallback path.' % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            f = getattr(l, name)      # AttributeError if the symbol is not exported
            f.restype = res
            f.argtypes = args
        # the ctypes mirrors against the structs the library was compiled with (include/vecvad_hip.h vv_abi_sizeof): a stale .so or a
        # mirror that missed a field would otherwise read garbage behind the shorter struct
        for which, cls in enumerate(ABI_STRUCTS):
            if l.vv_abi_sizeof(which) != C.sizeof(cls):
                raise VecVadHipError('%s: sizeof(%s) is %d in the library, %d in vec_vad_amd/_lib.py -- rebuild with '
                                     '`python -m vec_vad_amd.build`' % (LIB_PATH, cls.__name__, l.vv_abi_sizeof(which), C.sizeof(cls)))
        _lib = l
    return _lib


_STATUS = {1: 'VV_ERR_BAD_ARG', 2: 'VV_ERR_LAUNCH', 3: 'VV_ERR_UNSUPPORTED'}


def check(status, what=''):
    """status: the int every entry point returns (low byte = vv_status, bits 8.. = hipError_t of a failed launch)."""
    if status != 0:
        detail = ''
        if (status & 0xff) == 2 and _lib is not None:
            detail = ' (hipError %d: %s)' % (status >> 8, _lib.vv_status_string(status).decode())
        raise VecVadHipError('%s failed: %s%s' % (what or 'libvecvad_hip call', _STATUS.get(status & 0xff, status), detail))


def view(t, cstride, coff=0, gstride=0):
    """vv_view over a torch tensor (or raw pointer)."""
    ptr = t if isinstance(t, int) else t.data_ptr()
    return View(ptr, int(gstride), int(cstride), int(coff))


NULL_VIEW = View(None, 0, 0, 0)
