"""ctypes binding of libcpc_hip.so (the C ABI declared in include/cpc_hip.h).

The product path has NO fallback: if the library is missing or a launch is refused, this module raises.
Tensors are passed as raw device pointers; everything is launched on torch's current HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

F32, BF16 = 0, 1
GEMM_RELU, GEMM_OUT_F32, GEMM_TN_NO_TR, GEMM_FORCE_GENERIC, GEMM_SMALL_TILE, GEMM_NARROW_EPI, GEMM_NO_DMA, GEMM_SKIP_PAD_ROWS = 1, 2, 4, 8, 16, 32, 64, 128
GEMM_BIG_TILE = 0x100000          # CPC_GEMM_BIG_TILE: the 256 x 256 tile also where fewer than 200 of them exist
GEMM_LINEAR_K, GEMM_NO_PERS, GEMM_DIRECT_MASK, GEMM_KRANGE_EXACT = 256, 512, 1024, 2048

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcpc_hip.so")


class HipLibraryMissing(ImportError):
    pass


class GemmNTArgs(C.Structure):
    _fields_ = [("A", C.c_void_p), ("Bt", C.c_void_p), ("C", C.c_void_p), ("bias", C.c_void_p), ("mask", C.c_void_p),
                ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
                ("lda", C.c_longlong), ("ldb", C.c_longlong), ("ldc", C.c_longlong),
                ("a_rpi", C.c_int), ("a_item", C.c_longlong),
                ("b_rpi", C.c_int), ("b_item", C.c_longlong),
                ("c_rpi", C.c_int), ("c_item", C.c_longlong), ("c_valid", C.c_int),
                ("a_batch", C.c_longlong), ("b_batch", C.c_longlong), ("c_batch", C.c_longlong), ("batch", C.c_int),
                ("flags", C.c_int), ("dtype", C.c_int), ("a_extent", C.c_longlong), ("b_extent", C.c_longlong),
                ("a_rpi2", C.c_int), ("a_item2", C.c_longlong), ("c_rpi2", C.c_int), ("c_item2", C.c_longlong), ("k_ranges", C.c_void_p),
                ("k_taps", C.c_int), ("k_tap_stride", C.c_longlong), ("k_tap_stride_a", C.c_longlong)]


class ConvPrepJob(C.Structure):
    _fields_ = [("w", C.c_void_p), ("w_fwd", C.c_void_p), ("w_dgrad", C.c_void_p), ("Cout", C.c_int), ("Cin", C.c_int), ("kw", C.c_int),
                ("stride", C.c_int), ("D", C.c_int), ("tco", C.c_int), ("gx", C.c_int), ("first", C.c_int)]


class ConvPrepBatch:
    """cpc_conv_w_prep for a list of (w, w_fwd, w_dgrad, Cout, Cin, kw, stride) in one launch; the job table is planned on the host and
    uploaded once (tensor addresses are stable: parameters are views of the model's flat buffer, operands live as long as their engine)."""

    def __init__(self, jobs, device):
        n = len(jobs)
        arr = (ConvPrepJob * n)()
        for q, (w, fwd, dgrd, cout, cin, kw, stride) in zip(arr, jobs):
            q.w, q.w_fwd, q.w_dgrad, q.Cout, q.Cin, q.kw, q.stride = w.data_ptr(), fwd.data_ptr(), dgrd.data_ptr(), cout, cin, kw, stride
        tb, lds = C.c_int(0), C.c_int(0)
        _check(lib().cpc_conv_w_prep_plan(C.cast(arr, C.c_void_p), n, C.cast(C.byref(tb), C.c_void_p), C.cast(C.byref(lds), C.c_void_p)),
               "cpc_conv_w_prep_plan")
        self.n, self.blocks, self.lds = n, tb.value, lds.value
        self.table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)
        self.keep = [t for j in jobs for t in j[:3]]

    def run(self, dtype):
        call("cpc_conv_w_prep_batch", ptr(self.table), self.n, self.blocks, self.lds, dtype)


class GemmTNArgs(C.Structure):
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p),
                ("M", C.c_int), ("I", C.c_int), ("J", C.c_int),
                ("lda", C.c_longlong), ("ldb", C.c_longlong), ("ldc", C.c_longlong),
                ("a_rpi", C.c_int), ("a_item", C.c_longlong),
                ("b_rpi", C.c_int), ("b_item", C.c_longlong),
                ("a_batch", C.c_longlong), ("b_batch", C.c_longlong), ("c_batch", C.c_longlong), ("batch", C.c_int),
                ("nsplit", C.c_int), ("m_chunk", C.c_int), ("slab_stride", C.c_longlong),
                ("flags", C.c_int), ("dtype", C.c_int), ("c_rpi", C.c_int), ("c_item", C.c_longlong), ("a_rpi2", C.c_int), ("a_item2", C.c_longlong)]


_P, _I, _L, _F, _D, _U64, _U32 = C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_double, C.c_ulonglong, C.c_uint
_SIGNATURES = {
    "cpc_abi_version": ([], _I),
    "cpc_gemm_nt": ([C.POINTER(GemmNTArgs), _P], _I),
    "cpc_gemm_tn": ([C.POINTER(GemmTNArgs), _P], _I),
    "cpc_reduce_slabs": ([_P, _P, _I, _I, _I, _L, _I, _L, _L, _L, _P], _I),
    "cpc_colsum": ([_P, _P, _I, _I, _L, _I, _I, _P], _I),
    "cpc_conv1_fwd": ([_P, _P, _P, _P, _I, _I, _I, _I, _L, _I, _I, _I, _I, _P, _P], _I),
    "cpc_conv1_fwd_rows": ([_P, _P, _P, _P, _I, _I, _I, _I, _L, _I, _I, _I, _I, _P, _I, _I, _P], _I),
    "cpc_sign_bits": ([_P, _P, _L, _I, _P], _I),
    "cpc_conv1_bwd": ([_P, _P, _P, _I, _I, _I, _I, _L, _I, _I, _I, _I, _I, _P], _I),
    "cpc_reduce_conv_w": ([_P, _P, _I, _I, _I, _I, _L, _P], _I),
    "cpc_reduce_conv_w2d": ([_P, _P, _I, _L, _I, _I, _I, _I, _L, _L, _L, _I, _L, _P], _I),
    "cpc_conv_fwd": ([_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _L, _I, _P], _I),
    "cpc_conv_dgrad": ([_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _L, _I, _P, _P, _P], _I),
    "cpc_conv_dgrad_colsum_floats": ([_I, _I, _I, _I], _L),
    "cpc_conv_dgrad_conv1_floats": ([_I, _I, _I, _I, _I, _I], _L),
    "cpc_conv_dgrad_conv1": ([_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _L, _I, _I, _I, _L, _I, _P, _P], _I),
    "cpc_conv1_fused_reduce": ([_P, _P, _P, _P, _I, _I, _I, _I, _I, _P], _I),
    "cpc_conv_wgrad": ([_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _L, _I, _P], _I),
    "cpc_conv_dgrad_rows": ([_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _L, _I, _P, _P, _I, _I, _P], _I),
    "cpc_conv_dgrad_conv1_rows": ([_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _L, _I, _I, _I, _L, _I, _P, _I, _I, _P], _I),
    "cpc_conv1_fused_reduce_tiles": ([_P, _P, _P, _P, _I, _I, _I, _I, _P], _I),
    "cpc_conv_w_prep": ([_P, _P, _P, _I, _I, _I, _I, _I, _P], _I),
    "cpc_conv_w_prep_group": ([_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P], _I),
    "cpc_conv_w_prep_plan": ([_P, _I, _P, _P], _I),
    "cpc_conv_w_prep_batch": ([_P, _I, _I, _I, _I, _P], _I),
    "cpc_maxpool_fwd": ([_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P], _I),
    "cpc_maxpool_bwd": ([_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P], _I),
    "cpc_relu_row_bwd": ([_P, _P, _P, _I, _I, _L, _L, _I, _P], _I),
    "cpc_pe_scale_fwd": ([_P, _P, _P, _I, _I, _I, _L, _F, _I, _P], _I),
    "cpc_pe_scale_bwd": ([_P, _P, _P, _I, _I, _I, _L, _F, _I, _P], _I),
    "cpc_dropout": ([_P, _L, _F, _U64, _U32, _I, _P], _I),
    "cpc_dropout_mask": ([_P, _L, _F, _U64, _U32, _P], _I),
    "cpc_attn_fwd": ([_P, _P, _P, _I, _I, _I, _I, _F, _U64, _U32, _I, _P], _I),
    "cpc_attn_bwd": ([_P, _P, _P, _P, _I, _I, _I, _I, _F, _U64, _U32, _I, _P], _I),
    "cpc_add_ln_fwd": ([_P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _F, _U64, _U32, _I, _P], _I),
    "cpc_ln_tangent": ([_P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _U64, _U32, _P], _I),
    "cpc_ln_gp": ([_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _I, _F, _U64, _U32, _P], _I),
    "cpc_attn_tangent": ([_P, _P, _P, _P, _I, _I, _I, _I, _F, _U64, _U32, _P], _I),
    "cpc_attn_gp": ([_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _U64, _U32, _P], _I),
    "cpc_ln_bwd": ([_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _I, _P, _F, _U64, _U32, _I, _P], _I),
    "cpc_mean_time": ([_P, _P, _I, _I, _I, _I, _P], _I),
    "cpc_scalogram_pointwise": ([_P, _P, _P, _P, _I, _I, _I, _L, _I, _F, _F, _F, _F, _I, _I, _P], _I),
    "cpc_im2col2d": ([_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P], _I),
    "cpc_col2im2d": ([_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P], _I),
    "cpc_dw_fwd": ([_P, _P, _P, _L, _I, _I, _I, _I, _L, _I, _P], _I),
    "cpc_dw_bwd_col": ([_P, _P, _P, _L, _I, _I, _I, _I, _L, _I, _P], _I),
    "cpc_dw_bwd_w": ([_P, _P, _P, _L, _I, _I, _I, _I, _L, _I, _I, _P], _I),
    "cpc_bn_stats": ([_P, _P, _L, _I, _I, _I, _P], _I),
    "cpc_bn_finalize": ([_P, _I, _I, _D, _F, _F, _P, _P, _P, _P], _I),
    "cpc_bn_apply": ([_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P], _I),
    "cpc_bn_bwd_reduce": ([_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P], _I),
    "cpc_bn_bwd_apply": ([_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _D, _I, _I, _I, _I, _P], _I),
    "cpc_bn_apply_bits": ([_P, _P, _P, _P, _P, _P, _P, _I, _P, _I, _P], _I),
    "cpc_bn_apply_residual": ([_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _I, _P], _I),
    "cpc_bn_bwd_reduce_res": ([_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P], _I),
    "cpc_bn_bwd_apply_res": ([_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _D, _I, _P, _P, _I, _I, _I, _P], _I),
    "cpc_bn_bwd_reduce_bits": ([_P, _P, _P, _P, _P, _P, _P, _I, _I, _P], _I),
    "cpc_bn_bwd_apply_bits": ([_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _D, _I, _I, _P], _I),
    "cpc_maxpool2d_fwd": ([_P, _P, _P, _P, _I, _I, _I, _P], _I),
    "cpc_maxpool2d_bwd": ([_P, _P, _P, _P, _P, _I, _I, _I, _P], _I),
    "cpc_residual_add": ([_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P], _I),
    "cpc_residual_add_bwd": ([_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P], _I),
    "cpc_stem_supported": ([_I, _I, _I, _I, _I, _I, _I], _I),
    "cpc_stem_stats": ([_P, _P, _P, _P, _P, _P, _I, _P], _I),
    "cpc_stem_apply": ([_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P], _I),
    "cpc_stem_bwd_reduce": ([_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P], _I),
    "cpc_stem_bwd_wgrad": ([_P, _P, _P, _P, _P, _P, _P, _P, _P, _D, _P, _P, _P, _P, _I, _I, _P], _I),
    "cpc_stem_residual_add": ([_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P], _I),
    "cpc_stem_residual_bn_add": ([_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P], _I),
    "cpc_stem_residual_wgrad_bits": ([_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P], _I),
    "cpc_stem_residual_bwd": ([_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P], _I),
    "cpc_maxpool2d_select": ([_P, _P, _P, _P, _P, _I, _I, _I, _P], _I),
    "cpc_gp_direction": ([_P, _P, _L, _I, _F, _P, _I, _P], _I),
    "cpc_bn_gp_cross": ([_P, _P, _P, _P, _P, _P, _P, _I, _I, _P], _I),
    "cpc_relu_mask": ([_P, _P, _L, _I, _P], _I),
    "cpc_accumulate": ([_P, _P, _L, _I, _P], _I),
    "cpc_split3_bf16": ([_P, _P, _L, _P], _I),
    "cpc_cast2d": ([_P, _P, _I, _I, _L, _L, _I, _P], _I),
    "cpc_cast2d_batch": ([_P, _I, _I, _P], _I),
    "cpc_prep_frag": ([_P, _P, _I, _I, _L, _I, _I, _P], _I),
    "cpc_gru_tape_elems": ([_I, _I, _I, _I], _L),
    "cpc_gru_fwd": ([_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P], _I),
    "cpc_gru_fwd_h0": ([_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P], _I),
    "cpc_gru_bwd": ([_P, _P, _P, _P, _I, _I, _I, _I, _P], _I),
    "cpc_gru_gp_fwd": ([_P, _P, _P, _P, _P, _P, _I, _I, _I, _P], _I),
    "cpc_gru_gp_bwd": ([_P, _P, _P, _P, _I, _I, _I, _P], _I),
    "cpc_gru_set_streaming": ([_I], _I),
    "cpc_debug_set": ([_I, _I], _I),
    "cpc_nce_workspace_floats": ([_I, _I], _L),
    "cpc_nce_loss": ([_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _P], _I),
    "cpc_nce_all_workspace_floats": ([_I, _I], _L),
    "cpc_nce_loss_all": ([_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _P], _I),
    "cpc_score_lse": ([_P, _P, _P, _P, _P, _P, _I, _I, _I, _L, _L, _L, _I, _P], _I),
    "cpc_nce_lse_merge": ([_P, _P, _I, _I, _I, _F, _P, _P, _P], _I),
    "cpc_nce_fused_grad_blocks": ([_I, _I], _L),
    "cpc_nce_fused_grad": ([_P, _P, _P, _P, _P, _I, _I, _I, _L, _L, _I, _I, _F, _F, _F, _P], _I),
    "cpc_nce_fused_finalize": ([_P, _I, _P, _I, _P, _I, _P, _I, _F, _F, _I, _F, _I, _P, _P], _I),
    "cpc_gp_score_coeff": ([_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P], _I),
    "cpc_nce_eval_workspace_floats": ([_I, _I], _L),
    "cpc_nce_eval": ([_P, _P, _P, _I, _I, _I, _I, _I, _I, _P], _I),
    "cpc_adam": ([_P, _P, _P, _P, _L, _F, _F, _F, _F, _I, _F, _P, _P], _I),
    "cpc_adam_dev": ([_P, _P, _P, _P, _L, _F, _F, _F, _F, _P, _F, _P, _P], _I),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


def lib():
    """Loads libcpc_hip.so (once).  Raises HipLibraryMissing — never falls back to another implementation."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryMissing(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"(or `make -C {os.path.join(_HERE, 'csrc')}`); there is no CPU fallback for the CPC hot path")
        handle = C.CDLL(LIB_PATH)
        for name, (argtypes, restype) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.argtypes = argtypes
            fn.restype = restype
        if handle.cpc_abi_version() != 8:
            raise HipLibraryMissing("libcpc_hip.so ABI version mismatch; rebuild it")
        _lib = handle
    return _lib


class HipCallError(RuntimeError):
    pass


def _check(rc, what):
    if rc != 0:
        raise HipCallError(f"{what} failed with code {rc} ({'EINVAL: unsupported shape/argument' if rc == -22 else 'launch error'})")


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return F32
    if dt == torch.bfloat16:
        return BF16
    raise ValueError(f"unsupported storage dtype {dt}")


def ptr(t, offset_elems: int = 0):
    """Raw device address of a tensor's first element (+ offset in elements); None -> NULL."""
    if t is None:
        return None
    assert t.is_cuda, "HIP path needs device tensors"
    return C.c_void_p(t.data_ptr() + offset_elems * t.element_size())


# ----------------------------------------------------------------------------- optional per-kernel timing
class KernelTimer:
    """Brackets selected launches with HIP events on the launch stream (torch's current stream is the stream every
    kernel here is launched on).  ``only`` limits the bracketing to the named kernel symbols (None = all)."""

    def __init__(self, only=None, by_shape=False):
        self.only = set(only) if only is not None else None
        self.by_shape = by_shape
        self.records = {}
        # an event pair around a launch costs ~12 us of idle queue on MI355X (rocprofv3 kernel trace: 6 us per recorded event
        # between two back-to-back kernels); callers that time a long run switch ``active`` on for a sample of the steps only
        self.active = True

    def run(self, key, work, fn, shape=None):
        if not self.active or (self.only is not None and key not in self.only):
            return fn()
        if self.by_shape and shape is not None:
            key = f"{key} {shape}"
        start = torch.cuda.Event(enable_timing=True)
        end = torch.cuda.Event(enable_timing=True)
        start.record()
        fn()
        end.record()
        self.records.setdefault(key, []).append((start, end, work))

    def reset(self):
        self.records = {}

    def summary(self):
        """{key: (launches, total_ms, total_work)} — synchronises."""
        torch.cuda.synchronize()
        out = {}
        for key, recs in self.records.items():
            out[key] = (len(recs), sum(s.elapsed_time(e) for s, e, _ in recs), sum(w for _, _, w in recs))
        return out


_timer = None


def set_timer(timer):
    global _timer
    _timer = timer


def _variant(dtype, flags, tile=None):
    """Kernel-symbol label used by the timer: storage/output dtypes and, when known, the output tile (128 or 256)."""
    t = f",{tile}" if tile else ""
    if dtype == F32:
        return f"<f32,f32{t}>"
    return f"<bf16,f32{t}>" if flags & GEMM_OUT_F32 else f"<bf16,bf16{t}>"


def nt_tile(dtype, M, N, K, flags=0, batch=1):
    """Mirror of the launcher's tile choice for gemm_nt (csrc/gemm.hip launch_gemm_nt)."""
    ch = 8 if dtype == BF16 else 4
    fast = K % (8 * ch) == 0 and not (flags & GEMM_FORCE_GENERIC)
    big_tiles = ((M + 255) // 256) * ((N + 255) // 256) * batch
    return 256 if (fast and dtype == BF16 and N >= 256 and not (flags & GEMM_SMALL_TILE) and
                   ((M >= 1024 and big_tiles >= 200) or ((flags & GEMM_BIG_TILE) and M >= 256))) else 128


def tn_tile(dtype, M, I, J, nsplit, m_chunk, flags=0):
    """Mirror of the launcher's tile choice for gemm_tn (csrc/gemm.hip launch_gemm_tn)."""
    eff = m_chunk if nsplit > 1 else M
    fast = dtype == BF16 and not (flags & (GEMM_FORCE_GENERIC | GEMM_TN_NO_TR)) and I % 8 == 0 and I >= 8 and J >= 8
    return 256 if (fast and I >= 256 and J >= 256 and eff >= 1024 and not (flags & GEMM_SMALL_TILE)) else 128


# ----------------------------------------------------------------------------- thin call wrappers
def gemm_nt(A, Bt, Cout, M, N, K, lda, ldb, ldc, dtype, *, bias=None, mask=None, a_rpi=0, a_item=0, b_rpi=0, b_item=0,
            c_rpi=0, c_item=0, c_valid=0, a_batch=0, b_batch=0, c_batch=0, batch=1, flags=0, a_extent=0, b_extent=0,
            a_rpi2=0, a_item2=0, c_rpi2=0, c_item2=0, k_ranges=None, work=None, k_taps=0, k_tap_stride=0, k_tap_stride_a=0):
    """A, Bt, Cout, bias, mask are ctypes void pointers (see ptr()).  a_extent / b_extent: elements readable from A / Bt
    (0 = unchecked), see the over-read contract in include/cpc_hip.h."""
    args = GemmNTArgs(A, Bt, Cout, bias, mask, M, N, K, lda, ldb, ldc, a_rpi, a_item, b_rpi, b_item, c_rpi, c_item,
                      c_valid, a_batch, b_batch, c_batch, batch, flags, dtype, a_extent, b_extent, a_rpi2, a_item2, c_rpi2, c_item2, k_ranges,
                      k_taps, k_tap_stride, k_tap_stride_a)
    if _timer is not None:
        # (work: the FLOPs a launch with k_ranges executes, when the caller knows them)
        _timer.run("gemm_nt" + _variant(dtype, flags, nt_tile(dtype, M, N, K, flags, batch)), work if work is not None else 2.0 * M * N * K * batch,
                   lambda: _check(lib().cpc_gemm_nt(C.byref(args), stream_ptr()), "cpc_gemm_nt"), shape=(M, N, K, batch))
        return
    _check(lib().cpc_gemm_nt(C.byref(args), stream_ptr()), "cpc_gemm_nt")


def gemm_tn(A, B, Cout, M, I, J, lda, ldb, ldc, dtype, *, a_rpi=0, a_item=0, b_rpi=0, b_item=0, a_batch=0, b_batch=0,
            c_batch=0, batch=1, nsplit=1, m_chunk=0, slab_stride=0, flags=0, c_rpi=0, c_item=0, a_rpi2=0, a_item2=0):
    args = GemmTNArgs(A, B, Cout, M, I, J, lda, ldb, ldc, a_rpi, a_item, b_rpi, b_item, a_batch, b_batch, c_batch, batch,
                      nsplit, m_chunk, slab_stride, flags, dtype, c_rpi, c_item, a_rpi2, a_item2)
    if _timer is not None:
        _timer.run("gemm_tn" + _variant(dtype, flags if nsplit == 1 else flags | GEMM_OUT_F32,
                                        tn_tile(dtype, M, I, J, nsplit, m_chunk, flags)), 2.0 * M * I * J * batch,
                   lambda: _check(lib().cpc_gemm_tn(C.byref(args), stream_ptr()), "cpc_gemm_tn"), shape=(M, I, J, batch, nsplit))
        return
    _check(lib().cpc_gemm_tn(C.byref(args), stream_ptr()), "cpc_gemm_tn")


# Timing probes (tools/): CPC_PROBE_SKIP=<entry point>,... drops those launches, so the step's RESULTS ARE GARBAGE.  Honoured only
# together with CPC_ENABLE_PROBES=1, and announced on stderr; without the opt-in a stray CPC_PROBE_SKIP is refused loudly.
_PROBE_SKIP = frozenset(v for v in os.environ.get("CPC_PROBE_SKIP", "").split(",") if v)
PROBES_ENABLED = os.environ.get("CPC_ENABLE_PROBES", "0") == "1"
if _PROBE_SKIP:
    import sys as _sys
    if not PROBES_ENABLED:
        raise RuntimeError("CPC_PROBE_SKIP is set but CPC_ENABLE_PROBES=1 is not: timing probes drop kernel launches and make every "
                           "result wrong; unset CPC_PROBE_SKIP, or opt in explicitly for a timing run")
    print(f"[cpc_hip] WARNING: timing probe active, launches of {sorted(_PROBE_SKIP)} are SKIPPED -- results are garbage", file=_sys.stderr)


def call(name, *args, key=None, work=0.0, shape=None):
    """Generic call of an exported function; appends the current stream and checks the status.
    ``key`` / ``work``: kernel symbol and algorithmic FLOPs (or bytes) this launch is booked under by a KernelTimer."""
    if _PROBE_SKIP and name in _PROBE_SKIP:      # CPC_PROBE_SKIP=name,name: timing probes (the step's results are garbage)
        return
    fn = getattr(lib(), name)
    if _timer is not None:
        _timer.run(key or name, work, lambda: _check(fn(*args, stream_ptr()), name), shape=shape)
        return
    _check(fn(*args, stream_ptr()), name)
