"""ctypes binding of liblsps_hip.so (the C-ABI declared in include/lsps_hip.h).

There is NO fallback: if the shared library is missing or a tensor is not a contiguous float32
HIP tensor, the call raises.  The library is built in-tree by ``__graft_entry__.build()`` /
``make -C lsps_amd/csrc``.
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_long, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'liblsps_hip.so')        # options.hip_lib (LSPS_HIP_LIB) overrides it: kernel A/B experiments


class LspsOptions(ctypes.Structure):
    """include/lsps_hip.h: struct LspsOptions (the library's dispatch switches; -1 = the library's default)."""
    _fields_ = [('struct_size', c_int), ('wino4_split', c_int), ('fs2_cc', c_int), ('wino4w', c_int), ('wino4w_waves', c_int),
                ('chwn_group', c_int), ('c8w_queue', c_int), ('c8_stem_bf16', c_int), ('x3_plan', c_int), ('x3_ring', c_int)]


ACT_NONE, ACT_LRELU, ACT_TANH, ACT_SOFTPLUS = 0, 1, 2, 3
LOSS_L1, LOSS_L2, LOSS_SQ, LOSS_KLSD = 0, 1, 2, 3

_P = c_void_p
_SIGNATURES = {
    # name: (restype, argtypes)
    'lsps_version': (c_int, []),
    'lsps_last_error': (c_char_p, []),
    'lsps_device_cus': (c_int, []),
    'lsps_last_kernel': (c_char_p, [_P]),
    'lsps_set_math_mode': (c_int, [c_int]),
    'lsps_set_winograd': (c_int, [c_int]),
    'lsps_get_winograd': (c_int, []),
    'lsps_set_options': (c_int, [ctypes.POINTER(LspsOptions)]),
    'lsps_get_options': (c_int, [ctypes.POINTER(LspsOptions)]),
    'lsps_get_math_mode': (c_int, []),
    'lsps_pack_cache_begin': (c_int, [_P, c_size_t]),
    'lsps_pack_cache_end': (c_int, []),
    'lsps_pack_cache_frozen': (c_int, [_P, _P, _P, c_size_t, ctypes.c_ulonglong]),
    'lsps_conv2d_workspace_bytes': (c_size_t, [c_int] * 9),
    'lsps_conv2d_fwd': (c_int, [_P, _P, _P, _P] + [c_int] * 9 + [c_int, c_float, _P, c_size_t, _P]),
    'lsps_conv2d_in_fwd': (c_int, [_P, _P, _P, _P, _P] + [c_int] * 5 + [c_float, c_float, _P, c_size_t, _P]),
    'lsps_conv2d_in_fwd_nograd': (c_int, [_P, _P, _P, _P, _P] + [c_int] * 5 + [c_float, c_float, _P, c_size_t, _P]),
    'lsps_conv2d_dgrad': (c_int, [_P, _P, _P] + [c_int] * 9 + [_P, c_size_t, _P]),
    'lsps_conv2d_dgrad_acc': (c_int, [_P, _P, _P, _P] + [c_int] * 9 + [_P, c_size_t, _P]),
    'lsps_conv2d_dgrad_inbwd': (c_int, [_P, _P, _P, _P, _P] + [c_int] * 5 + [c_float, _P, c_size_t, _P]),
    'lsps_conv2d_wgrad': (c_int, [_P, _P, _P, _P] + [c_int] * 9 + [_P, c_size_t, _P]),
    'lsps_conv2d_grouped_fwd': (c_int, [_P, _P, _P, _P] + [c_int] * 10 + [c_int, c_float, _P, c_size_t, _P]),
    'lsps_conv2d_grouped_dgrad': (c_int, [_P, _P, _P] + [c_int] * 10 + [_P, c_size_t, _P]),
    'lsps_conv2d_grouped_wgrad': (c_int, [_P, _P, _P, _P] + [c_int] * 10 + [_P, c_size_t, _P]),
    'lsps_transpose2d': (c_int, [_P, _P, c_long, c_long, _P]),
    'lsps_conv3x3s2_chwn_workspace_bytes': (c_size_t, [c_int] * 5),
    'lsps_conv3x3s2_chwn_fwd': (c_int, [_P, _P, _P, _P] + [c_int] * 5 + [c_int, c_float, _P, c_size_t, _P]),
    'lsps_conv3x3s2_chwn_dgrad': (c_int, [_P, _P, _P] + [c_int] * 5 + [_P, c_size_t, _P]),
    'lsps_conv3x3s2_chwn_wgrad': (c_int, [_P, _P, _P] + [c_int] * 5 + [_P, c_size_t, _P]),
    'lsps_convT2d_workspace_bytes': (c_size_t, [c_int] * 10),
    'lsps_convT2d_fwd': (c_int, [_P, _P, _P, _P] + [c_int] * 10 + [c_int, c_float, _P, c_size_t, _P]),
    'lsps_convT2d_dgrad': (c_int, [_P, _P, _P] + [c_int] * 10 + [_P, c_size_t, _P]),
    'lsps_convT2d_wgrad': (c_int, [_P, _P, _P, _P] + [c_int] * 10 + [_P, c_size_t, _P]),
    'lsps_inorm_fwd': (c_int, [_P, _P, _P, _P, c_int, c_int, c_float, c_float, _P]),
    'lsps_inorm_bwd': (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_float, _P]),
    'lsps_act_bwd': (c_int, [_P, _P, _P, c_long, c_int, c_float, _P]),
    'lsps_act_bwd_bias': (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, _P, c_size_t, _P]),
    'lsps_loss_workspace_bytes': (c_size_t, [c_long]),
    'lsps_loss_fwd': (c_int, [c_int, _P, _P, c_long, c_float, _P, _P, c_size_t, _P]),
    'lsps_loss_bwd': (c_int, [c_int, _P, _P, c_long, c_float, _P, _P, _P, _P]),
    'lsps_bce_sigmoid_fwd': (c_int, [_P, c_long, c_float, _P, _P, c_size_t, _P]),
    'lsps_bce_sigmoid_bwd': (c_int, [_P, c_long, c_float, _P, _P, _P]),
    'lsps_linear_fwd': (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, _P]),
    'lsps_linear_bwd': (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, _P, _P]),
    'lsps_adam_step': (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int] + [c_float] * 6 + [_P]),
    'lsps_axpy': (c_int, [_P, _P, c_float, _P, c_long, _P]),
    'lsps_bnorm_workspace_bytes': (c_size_t, [c_int]),
    'lsps_bnorm_fwd': (c_int, [_P] * 8 + [c_int] * 4 + [c_float] * 3 + [_P, c_size_t, _P]),
    'lsps_bnorm_bwd': (c_int, [_P] * 8 + [c_int] * 4 + [_P, c_size_t, _P]),
    'lsps_act_fwd': (c_int, [_P, _P, c_long, c_int, c_float, _P]),
    'lsps_mul_add': (c_int, [_P, _P, _P, _P, c_long, _P]),
    'lsps_c8_conv3x3_ok': (c_int, [c_int] * 5),
    'lsps_c8_conv3x3_workspace_bytes': (c_size_t, [c_int] * 2),
    'lsps_c8_from_nchw': (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    'lsps_c8_to_nchw': (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    'lsps_c8_add': (c_int, [_P, _P, _P, c_long, _P]),
    'lsps_c8_add_nchw': (c_int, [_P, _P, _P, c_int, c_int, c_int, _P]),
    'lsps_c8_conv3x3_fwd': (c_int, [_P, _P, _P, _P] + [c_int] * 5 + [_P, c_size_t, _P]),
    'lsps_c8_conv3x3_in_fwd': (c_int, [_P, _P, _P, _P, _P] + [c_int] * 5 + [c_float, c_float, _P, c_size_t, _P]),
    'lsps_c8_conv3x3_dgrad_acc': (c_int, [_P, _P, _P, _P] + [c_int] * 5 + [_P, c_size_t, _P]),
    'lsps_c8_conv3x3_dgrad_inbwd': (c_int, [_P, _P, _P, _P, _P] + [c_int] * 5 + [c_float, _P, c_size_t, _P]),
    'lsps_c8_conv3x3_wgrad_workspace_bytes': (c_size_t, [c_int] * 3),
    'lsps_c8_conv3x3_wgrad': (c_int, [_P, _P, _P] + [c_int] * 5 + [_P, c_size_t, _P]),
    'lsps_c8_inorm_bwd': (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_float, _P]),
    'lsps_c8_conv3x3s2_ok': (c_int, [c_int] * 5),
    'lsps_c8_convT3x3s2_ok': (c_int, [c_int] * 5),
    'lsps_c8_conv3x3s2_workspace_bytes': (c_size_t, [c_int] * 5),
    'lsps_c8_conv3x3s2_fwd': (c_int, [_P, _P, _P, _P] + [c_int] * 5 + [c_float, _P, c_size_t, _P]),
    'lsps_c8_conv3x3s2_dgrad': (c_int, [_P, _P, _P] + [c_int] * 5 + [_P, c_size_t, _P]),
    'lsps_c8_conv3x3s2_wgrad': (c_int, [_P, _P, _P] + [c_int] * 5 + [_P, c_size_t, _P]),
    'lsps_c8_convT3x3s2_fwd': (c_int, [_P, _P, _P, _P] + [c_int] * 5 + [c_float, _P, c_size_t, _P]),
    'lsps_c8_convT3x3s2_dgrad': (c_int, [_P, _P, _P] + [c_int] * 5 + [_P, c_size_t, _P]),
    'lsps_c8_convT3x3s2_wgrad': (c_int, [_P, _P, _P] + [c_int] * 5 + [_P, c_size_t, _P]),
    'lsps_c8_act_bwd_bias_workspace_bytes': (c_size_t, [c_int] * 2),
    'lsps_c8_act_bwd_bias': (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_float, _P, c_size_t, _P]),
    'lsps_c8_stem_ok': (c_int, [c_int] * 8),
    'lsps_c8_stem_workspace_bytes': (c_size_t, [c_int] * 3),
    'lsps_c8_stem_fwd': (c_int, [_P, _P, _P, _P] + [c_int] * 8 + [c_float, _P]),
    'lsps_c8_stem_wgrad': (c_int, [_P, _P, _P, _P, _P] + [c_int] * 8 + [c_float, _P, c_size_t, _P]),
    'lsps_c8_stem_dgrad_ok': (c_int, [c_int] * 8),
    'lsps_c8_stem_dgrad': (c_int, [_P, _P, _P, _P] + [c_int] * 8 + [c_float, _P]),
    'lsps_c8_pw1_workspace_bytes': (c_size_t, [c_int] * 2),
    'lsps_c8_pw1_fwd': (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, _P]),
    'lsps_c8_pw1_dgrad': (c_int, [_P, _P, _P, c_int, c_int, c_int, _P]),
    'lsps_c8_pw1_wgrad': (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P, c_size_t, _P]),
    'lsps_c8_conv3x3s2_dgrad_act': (c_int, [_P, _P, _P, c_float, _P, _P] + [c_int] * 5 + [_P, c_size_t, _P]),
    'lsps_c8_convT3x3s2_dgrad_act': (c_int, [_P, _P, _P, c_float, _P, _P] + [c_int] * 5 + [_P, c_size_t, _P]),
    'lsps_c8_pw1_dgrad_act_workspace_bytes': (c_size_t, [c_int] * 2),
    'lsps_c8_pw1_dgrad_act': (c_int, [_P, _P, _P, c_float, _P, _P, _P, _P, c_int, c_int, c_int, _P, c_size_t, _P]),
    'lsps_conv2d_stem_wgrad_act_ok': (c_int, [c_int] * 8),
    'lsps_conv2d_stem_wgrad_act': (c_int, [_P, _P, _P, _P, _P] + [c_int] * 8 + [c_float, _P, c_size_t, _P]),
    'lsps_pw1_dgrad_act_workspace_bytes': (c_size_t, [c_int] * 2),
    'lsps_pw1_dgrad_act': (c_int, [_P, _P, _P, c_float, _P, _P, _P, _P, c_int, c_int, c_int, _P, c_size_t, _P]),
    'lsps_x3_split_nchw': (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    'lsps_x3_join_nchw': (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    'lsps_x3_conv3x3s2_ok': (c_int, [c_int] * 5),
    'lsps_x3_conv3x3s2_workspace_bytes': (c_size_t, [c_int] * 5),
    'lsps_x3_conv3x3s2_plan': (c_int, [c_int] * 6 + [ctypes.POINTER(c_int)]),
    'lsps_x3_conv3x3s2_fwd': (c_int, [_P, _P, _P, _P, _P] + [c_int] * 5 + [c_float, _P, c_size_t, _P]),
    'lsps_x3_conv3x3s2_dgrad': (c_int, [_P, _P, _P, _P, _P, c_float, _P] + [c_int] * 5 + [_P, c_size_t, _P]),
    'lsps_x3_conv3x3s2_wgrad': (c_int, [_P, _P, _P] + [c_int] * 5 + [_P, c_size_t, _P]),
    'lsps_x3_convT3x3s2_ok': (c_int, [c_int] * 5),
    'lsps_x3_convT3x3s2_fwd': (c_int, [_P, _P, _P, _P, _P] + [c_int] * 5 + [c_float, _P, c_size_t, _P]),
    'lsps_x3_convT3x3s2_dgrad': (c_int, [_P, _P, _P, _P, _P, c_float, _P] + [c_int] * 5 + [_P, c_size_t, _P]),
    'lsps_x3_convT3x3s2_wgrad': (c_int, [_P, _P, _P] + [c_int] * 5 + [_P, c_size_t, _P]),
    'lsps_x3_act_bwd_bias_workspace_bytes': (c_size_t, [c_int] * 2),
    'lsps_x3_act_bwd_bias': (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_float, _P, c_size_t, _P]),
    'lsps_pw1_dgrad_act_x3': (c_int, [_P, _P, _P, c_float, _P, _P, _P, _P, c_int, c_int, c_int, _P, c_size_t, _P]),
    'lsps_x3_stem_ok': (c_int, [c_int] * 8),
    'lsps_x3_stem_fwd': (c_int, [_P, _P, _P, _P] + [c_int] * 8 + [c_float, _P]),
    'lsps_crop_normalize': (c_int, [_P, _P, _P, _P, c_int, c_int, _P]),
    'lsps_crop_augment': (c_int, [_P, _P, _P, c_int, c_int, c_int, _P]),
}
EXPORTS = tuple(sorted(_SIGNATURES))

_lib = None


class LspsHipError(RuntimeError):
    pass


def lib():
    """Loads (once) and returns the ctypes handle; raises if the library has not been built."""
    global _lib
    if _lib is None:
        # torch must load ITS libamdhip64 first: liblsps_hip.so then binds to the same HIP runtime
        # (two runtimes in one process => "no ROCm-capable device" on the second one).
        import torch  # noqa: F401
        from . import options
        path = options.get().hip_lib or LIB_PATH
        if not os.path.exists(path):
            raise LspsHipError("liblsps_hip.so not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'`"
                               % path)
        h = ctypes.CDLL(path)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(h, name)           # AttributeError if a declared symbol is not exported
            fn.restype, fn.argtypes = res, args
        _lib = h
        # the library reads no environment variable: its switches come from the options object, at load and on every change
        push_options(options.get())
        rc = h.lsps_set_winograd(int(options.get().wino))
        if rc != 0:
            raise LspsHipError("LSPS_WINO=%r: %s" % (options.get().wino, h.lsps_last_error().decode()))
    return _lib


def csrc_sha16():
    """sha256[:16] over the library's sources (csrc/*.hip, *.h, Makefile, include/lsps_hip.h): names the BUILD a measurement file
    (profiles/r*_traffic*.json) belongs to; bench.py refuses PMC numbers taken on other sources."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(_HERE, 'csrc', '*.hip')) + glob.glob(os.path.join(_HERE, 'csrc', '*.h')))
    files += [os.path.join(_HERE, 'csrc', 'Makefile'), os.path.join(os.path.dirname(_HERE), 'include', 'lsps_hip.h')]
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def push_options(opt):
    """lsps_set_options with the library's share of `opt` (lsps_amd/options.py: Options.native())."""
    blk = LspsOptions(struct_size=ctypes.sizeof(LspsOptions), **opt.native())
    rc = _lib.lsps_set_options(ctypes.byref(blk))
    if rc != 0:
        msg = _lib.lsps_last_error()
        raise LspsHipError("lsps_set_options failed (rc=%d): %s" % (rc, msg.decode() if msg else ''))


def native_options():
    """What the library reports as in force (lsps_get_options): {field: int}."""
    blk = LspsOptions()
    check(lib().lsps_get_options(ctypes.byref(blk)), 'lsps_get_options')
    return dict((f, int(getattr(blk, f))) for f, _ in LspsOptions._fields_ if f != 'struct_size')


def check(rc, what):
    if rc != 0:
        msg = lib().lsps_last_error()
        raise LspsHipError("%s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else ''))


# ------------------------------------------------------------------------------------------
# tensor plumbing (torch is used for device memory + streams only)
# ------------------------------------------------------------------------------------------
def ptr(t, dtype=None):
    """Device pointer of a contiguous HIP tensor of `dtype` (default float32), or None -> NULL."""
    if t is None:
        return None
    import torch
    dtype = dtype or torch.float32
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise LspsHipError("lsps_amd ops need HIP device tensors (got %s); there is no CPU fallback"
                           % (t.device if hasattr(t, 'device') else type(t)))
    if t.dtype != dtype or not t.is_contiguous():
        raise LspsHipError("lsps_amd ops need contiguous %s tensors (dtype=%s contiguous=%s)"
                           % (dtype, t.dtype, t.is_contiguous()))
    return t.data_ptr()


def stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


_workspaces = {}
_retired = []          # buffers outgrown DURING a stream capture: the captured kernels keep their addresses


def workspace(nbytes, device):
    """One scratch buffer per (device, stream), grown on demand: ops on one stream run in order, so they share it; a second
    stream (the trainer overlaps independent branches of a step, a hipGraph capture runs on its own stream) gets its own.
    While the stream is being captured nothing may synchronise: the outgrown buffer is kept alive instead (its address is
    baked into the kernels captured so far) and a fresh stream starts at the size of the largest buffer of the device."""
    import torch
    cur = torch.cuda.current_stream(device)
    key = (device.type, device.index, cur.cuda_stream)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        capturing = torch.cuda.is_current_stream_capturing()
        if buf is not None:
            if capturing:
                _retired.append(buf)
            else:
                cur.synchronize()                       # outstanding users of the old buffer
        nbytes = max(int(nbytes * 1.25), 1 << 20)
        if buf is None:
            nbytes = max([nbytes] + [b.numel() for k, b in _workspaces.items() if k[:2] == key[:2]])
        buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf.data_ptr(), buf.numel()
