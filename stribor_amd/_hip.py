"""ctypes binding of libstribor_hip.so (the C ABI declared in include/stribor_hip.h).

The product path has NO fallback: if the shared library is missing or a tensor is not on a
ROCm device, the call raises.  Build the library with ``python -c "import __graft_entry__ as g; g.build()"``
or ``make -C stribor_amd/csrc -j8``.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('STRIBOR_HIP_LIB', os.path.join(_HERE, 'libstribor_hip.so'))   # override: experiments only

SX_F32, SX_BF16 = 0, 1
SX_MAX_STEPS = 128
SX_ABI_VERSION = 3
GEMM_F32, GEMM_F16X3 = 0, 1
FLAG_RQS_NEG_DISCRIMINANT, FLAG_NONFINITE, FLAG_F16_RANGE = 1, 2, 4

STEP_COUPLING_AFFINE = 1
STEP_AFFINE_CONST = 2
STEP_LINEAR_TILE = 3
STEP_MLP_HIDDEN = 5
STEP_MLP_HIDDEN2 = 6
STEP_MLP_OUT_TILE = 7
STEP_ROW_SCALE_EXP = 9
STEP_RQS_HIDDEN = 10
STEP_RQS_PHASE = 11
STEP_COUPLING_AFFINE_BWD = 12
STEP_CPL_HIDDEN = 13
STEP_CPL_HIDDEN2 = 14
STEP_COUPLING_AFFINE_DEEP = 15
STEP_COUPLING_AFFINE_BWD_A = 16
STEP_COUPLING_AFFINE_BWD_B = 17
STEP_LINEAR_BWD = 18
STEP_POINTWISE = 19
STEP_COUPLING_TIME = 20
STEP_COUPLING_AFFINE_HC = 21
STEP_WIDE_HIDDEN = 22
STEP_WIDE_AFFINE_TILE = 23
STEP_MLP_INPUT = 24

WGRAD_ROW_MAJOR, WGRAD_ROW_GROUPS, WGRAD_ROW_GROUPS_F16X3 = 0, 1, 3

ACT_TANH_FOLDED = 9
ACT_CODES = {'Identity': 0, 'Tanh': 1, 'ReLU': 2, 'Sigmoid': 3, 'ELU': 4, 'Softplus': 5, 'LeakyReLU': 6,
             'SiLU': 7, 'GELU': 8}

# every symbol include/stribor_hip.h declares (tests check that the library exports all of them)
EXPORTS = ['sx_abi_version', 'sx_fragment_mode', 'sx_last_error', 'sx_build_id', 'sx_absmax2', 'sx_permute', 'sx_affine_coupling', 'sx_rqs_coupling',
           'sx_cubic_coupling', 'sx_pointwise', 'sx_rqs_inverse_bwd', 'sx_rqs_forward_bwd', 'sx_affine_coupling_bwd', 'sx_time_affine_coupling', 'sx_cubic_inverse_bwd', 'sx_cubic_forward_bwd', 'sx_pointwise_bwd',
           'sx_unit_normal_logprob', 'sx_sum_f64', 'sx_packed_linear_floats', 'sx_pack_linear', 'sx_pack_linear_bound', 'sx_pack_linear_batch', 'sx_flow_run', 'sx_flow_run2', 'sx_flow_redo_words',
           'sx_flow_launch_info', 'sx_wgrad', 'sx_wgrad_layer', 'sx_colsum', 'sx_tri_inverse_f64',
           'sx_wgrad_scratch_floats', 'sx_wgrad_layer_scratch_floats', 'sx_flow_bwd_max_steps', 'sx_flow_bwd_partials',
           'sx_flow_bwd_run', 'sx_wgrad_reduce', 'sx_wgrad_reduce_batch', 'sx_rqs_slab_slots', 'sx_rqs_slab_scratch_floats', 'sx_rqs_slab_bwd', 'sx_rqs_slab_l1_scratch_floats', 'sx_rqs_slab_l1_bwd',
           'sx_rqs_slab_fwd_scratch_floats', 'sx_rqs_slab_fwd', 'sx_rqs_slab_hidden_floats', 'sx_rqs_slab_hidden']


class HipLibraryMissing(RuntimeError):
    pass


class sx_reduce_job(C.Structure):
    _fields_ = [('part_off', C.c_int64), ('dW_off', C.c_int64), ('db_off', C.c_int64), ('ldw', C.c_int64), ('row_map', C.c_void_p),
                ('col_map', C.c_void_p), ('M32', C.c_int32), ('N32', C.c_int32), ('m_valid', C.c_int32), ('n_valid', C.c_int32)]


class sx_pack_job(C.Structure):
    _fields_ = [('W', C.c_void_p), ('b', C.c_void_p), ('row_idx', C.c_void_p), ('col_idx', C.c_void_p), ('row_scale', C.c_void_p),
                ('bias_scale', C.c_void_p), ('dst', C.c_void_p), ('bound_out', C.c_void_p), ('out_dim', C.c_int32),
                ('in_dim', C.c_int32), ('m_tiles', C.c_int32), ('k_tiles', C.c_int32), ('transpose', C.c_int32),
                ('fold_ones', C.c_float)]


class sx_step(C.Structure):
    _fields_ = [('kind', C.c_int32), ('c0', C.c_int32), ('ct', C.c_int32), ('t0', C.c_int32), ('tt', C.c_int32),
                ('reverse', C.c_int32), ('act', C.c_int32), ('blob_off', C.c_uint32), ('blob_floats', C.c_uint32),
                ('ldj_scale', C.c_float), ('ldj_const', C.c_float), ('pad_', C.c_int32)]


class sx_program(C.Structure):
    _fields_ = [('n_steps', C.c_int32), ('dim', C.c_int32), ('latent_dim', C.c_int32), ('x_tiles', C.c_int32),
                ('tiles', C.c_int32), ('h_tiles', C.c_int32), ('identity_cols', C.c_int32), ('pad_', C.c_int32),
                ('steps', sx_step * SX_MAX_STEPS)]


_lib: Optional[C.CDLL] = None


def _declare(lib: C.CDLL) -> None:
    vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    lib.sx_abi_version.restype = i32
    lib.sx_abi_version.argtypes = []
    lib.sx_fragment_mode.restype = i32
    lib.sx_fragment_mode.argtypes = []
    lib.sx_last_error.restype = C.c_char_p
    lib.sx_last_error.argtypes = []
    lib.sx_build_id.restype = C.c_char_p
    lib.sx_build_id.argtypes = []
    lib.sx_permute.restype = i32
    lib.sx_permute.argtypes = [vp, vp, vp, i64, i32, i32, vp]
    lib.sx_affine_coupling.restype = i32
    lib.sx_affine_coupling.argtypes = [vp, vp, vp, vp, i64, vp, i32, i32, i64, i32, i32, i32, i32, f32, vp]
    lib.sx_rqs_coupling.restype = i32
    lib.sx_rqs_coupling.argtypes = [vp, vp, vp, vp, vp, i64, vp, i32, i32, i32, f32, f32, f32, f32, i64, i32, i32,
                                    i32, i32, f32, vp, vp]
    lib.sx_cubic_coupling.restype = i32
    lib.sx_cubic_coupling.argtypes = [vp, vp, vp, vp, vp, i64, vp, i32, i32, i32, f32, f32, i64, i32, i32, i32, i32, f32, vp]
    lib.sx_rqs_inverse_bwd.restype = i32
    lib.sx_rqs_inverse_bwd.argtypes = [vp, vp, vp, vp, vp, i64, vp, vp, vp, i32, i32, i32, f32, f32, f32, f32, i64, i32, f32, vp]
    lib.sx_rqs_slab_slots.restype = i32
    lib.sx_rqs_slab_slots.argtypes = [i32]
    lib.sx_rqs_slab_scratch_floats.restype = C.c_size_t
    lib.sx_rqs_slab_scratch_floats.argtypes = [i64, i32, i32]
    lib.sx_rqs_slab_bwd.restype = i32
    lib.sx_rqs_slab_bwd.argtypes = [vp, vp, vp, vp, vp, i64, i32, vp, vp, vp, vp, vp, i64, vp, i64, vp, vp, i32, i32, i32, f32, f32, f32,
                                    f32, i64, i32, f32, i32, vp, vp, vp, vp]
    lib.sx_rqs_slab_l1_scratch_floats.restype = C.c_size_t
    lib.sx_rqs_slab_l1_scratch_floats.argtypes = [i32, i32]
    lib.sx_rqs_slab_l1_bwd.restype = i32
    lib.sx_rqs_slab_l1_bwd.argtypes = [vp, vp, i64, i32, vp, vp, vp, vp, vp, vp, i64, vp, vp, i32, i64, i32, vp, vp, vp, vp]
    lib.sx_rqs_slab_fwd_scratch_floats.restype = C.c_size_t
    lib.sx_rqs_slab_fwd_scratch_floats.argtypes = [i64, i32]
    lib.sx_rqs_slab_fwd.restype = i32
    lib.sx_rqs_slab_fwd.argtypes = [vp, vp, i64, i32, vp, vp, vp, vp, i32, i32, vp, i32, i32, f32, f32, f32, f32, i64, i32, i32, f32, i32, i32, i32, vp, vp, vp]
    lib.sx_rqs_slab_hidden_floats.restype = C.c_size_t
    lib.sx_rqs_slab_hidden_floats.argtypes = [i64, i32]
    lib.sx_rqs_slab_hidden.restype = i32
    lib.sx_rqs_slab_hidden.argtypes = [vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, vp, vp]
    lib.sx_rqs_forward_bwd.restype = i32
    lib.sx_rqs_forward_bwd.argtypes = lib.sx_rqs_inverse_bwd.argtypes
    lib.sx_affine_coupling_bwd.restype = i32
    lib.sx_affine_coupling_bwd.argtypes = [vp, vp, vp, vp, i64, vp, vp, vp, i32, i32, i64, i32, i32, f32, vp]
    lib.sx_time_affine_coupling.restype = i32
    lib.sx_time_affine_coupling.argtypes = [vp, vp, vp, vp, i64, vp, vp, i32, vp, i32, i32, i64, i32, i32, i32, i32, f32, vp]
    lib.sx_cubic_inverse_bwd.restype = i32
    lib.sx_cubic_inverse_bwd.argtypes = [vp, vp, vp, vp, vp, vp, i64, vp, vp, vp, i32, i32, i32, f32, f32, i64, i32, f32, vp]
    lib.sx_cubic_forward_bwd.restype = i32
    lib.sx_cubic_forward_bwd.argtypes = [vp, vp, vp, vp, vp, i64, vp, vp, vp, i32, i32, i32, f32, f32, i64, i32, f32, vp]
    lib.sx_pointwise_bwd.restype = i32
    lib.sx_pointwise_bwd.argtypes = [vp, vp, vp, vp, vp, i64, i32, i32, f32, vp]
    lib.sx_pointwise.restype = i32
    lib.sx_pointwise.argtypes = [vp, vp, vp, vp, i64, i32, i32, i32, f32, i32, vp]
    lib.sx_absmax2.restype = i32
    lib.sx_absmax2.argtypes = [vp, i64, vp, i64, vp, vp]
    lib.sx_unit_normal_logprob.restype = i32
    lib.sx_unit_normal_logprob.argtypes = [vp, vp, vp, i64, i32, i32, vp]
    lib.sx_sum_f64.restype = i32
    lib.sx_sum_f64.argtypes = [vp, i64, vp, vp]
    lib.sx_packed_linear_floats.restype = C.c_size_t
    lib.sx_packed_linear_floats.argtypes = [i32, i32]
    lib.sx_pack_linear.restype = i32
    lib.sx_pack_linear.argtypes = [vp, vp, i32, i32, vp, vp, i32, i32, vp, vp, f32, i32, i32, vp, vp, vp]
    lib.sx_pack_linear_bound.restype = i32
    lib.sx_pack_linear_bound.argtypes = [vp, vp, i32, i32, vp, vp, i32, i32, vp, vp, f32, i32, i32, vp, vp, vp, vp]
    lib.sx_pack_linear_batch.restype = i32
    lib.sx_pack_linear_batch.argtypes = [vp, i32, i32, i32, vp, vp]
    lib.sx_wgrad_reduce_batch.restype = i32
    lib.sx_wgrad_reduce_batch.argtypes = [vp, vp, vp, i32, i32, i32, vp]
    lib.sx_flow_run.restype = i32
    lib.sx_flow_run.argtypes = [C.POINTER(sx_program), vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, vp, vp, i64, i32, i32, vp, vp, vp]
    lib.sx_flow_run2.restype = i32
    lib.sx_flow_run2.argtypes = [C.POINTER(sx_program), vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, vp, vp, i64, i32, i32, vp, vp, vp]
    lib.sx_flow_redo_words.restype = C.c_size_t
    lib.sx_flow_redo_words.argtypes = [i64]
    lib.sx_wgrad.restype = i32
    lib.sx_wgrad.argtypes = [vp, i64, i32, vp, i64, i32, i64, i32, vp, i64, vp, vp, vp, vp, vp]
    lib.sx_wgrad_scratch_floats.restype = C.c_size_t
    lib.sx_wgrad_scratch_floats.argtypes = [i32, i32, i32]
    lib.sx_wgrad_layer_scratch_floats.restype = C.c_size_t
    lib.sx_wgrad_layer_scratch_floats.argtypes = [i32, i32, i32]
    lib.sx_wgrad_layer.restype = i32
    lib.sx_wgrad_layer.argtypes = [vp, i64, i64, i32, i32, i32, i32, vp, i64, vp, vp, vp, i64, vp, vp, vp, vp]
    lib.sx_colsum.restype = i32
    lib.sx_colsum.argtypes = [vp, i64, i64, i32, vp, vp, vp]
    lib.sx_tri_inverse_f64.restype = i32
    lib.sx_tri_inverse_f64.argtypes = [vp, vp, i32, i32, i32, i32, vp]
    lib.sx_flow_bwd_max_steps.restype = i32
    lib.sx_flow_bwd_max_steps.argtypes = []
    lib.sx_flow_bwd_partials.restype = i32
    lib.sx_flow_bwd_partials.argtypes = [C.POINTER(sx_program), i64, C.POINTER(i32), C.POINTER(i64)]
    lib.sx_flow_bwd_run.restype = i32
    lib.sx_flow_bwd_run.argtypes = [C.POINTER(sx_program), vp, vp, vp, vp, vp, vp, vp, i64, vp, vp, vp]
    lib.sx_wgrad_reduce.restype = i32
    lib.sx_wgrad_reduce.argtypes = [vp, i32, i32, i32, vp, i64, vp, i32, i32, vp, vp, vp]
    lib.sx_flow_launch_info.restype = i32
    lib.sx_flow_launch_info.argtypes = [C.POINTER(sx_program), i64, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]


def lib() -> C.CDLL:
    """The loaded library; raises HipLibraryMissing (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryMissing(
                f'{LIB_PATH} not found: stribor_amd has no CPU fallback. Build it with '
                f'`make -C {os.path.join(_HERE, "csrc")} -j8` (hipcc, --offload-arch=gfx950).')
        l = C.CDLL(LIB_PATH)
        _declare(l)
        if l.sx_abi_version() != SX_ABI_VERSION:
            raise HipLibraryMissing(f'{LIB_PATH}: ABI version {l.sx_abi_version()} != {SX_ABI_VERSION}; rebuild')
        _lib = l
    return _lib


def build_id() -> str:
    """sha256[:16] of the sources the loaded library was built from (Makefile rule sx_build_id.inc)."""
    return lib().sx_build_id().decode()


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().sx_last_error().decode(errors='replace')
        raise RuntimeError(f'{what} failed (rc={rc}): {msg}')


def require_device(t: torch.Tensor, name: str = 'input') -> None:
    if not t.is_cuda:
        raise RuntimeError(f'stribor_amd: {name} must live on a ROCm device (got {t.device}); this package runs '
                           f'the coupling-flow path on MI355X HIP kernels only and has no CPU fallback.')


def dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return SX_F32
    if t.dtype == torch.bfloat16:
        return SX_BF16
    raise TypeError(f'stribor_amd: unsupported storage dtype {t.dtype} (float32 or bfloat16)')


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_cur_device = getattr(torch._C, '_cuda_getDevice', None)


def stream() -> int:
    """The hipStream_t of torch's current stream on the current device.  torch.cuda.current_stream() costs ~8 us of a
    17 us small-batch call (device-index resolution in Python); the raw accessor is one C call."""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


# ---- GEMM arithmetic of the MFMA path ---------------------------------------------------------------------------------
# 'fast'  fp16 x 3 split on the matrix pipe (fp32-grade: ~2^-22 per product).  A sample whose flow state or Tanh-conditioner
#         input leaves fp16's range (|v| > 65504) is rescaled by an exact power of two inside the kernel (round 5).
#         What still must stay within the range: WEIGHTS, hidden activations of non-Tanh coupling conditioners, operands of the
#         training backward -- such a sample comes back as NaN, never as a plausible number, and the next call (or
#         check_errors()) raises GemmRangeError.
# 'exact' v_mfma_f32_32x32x2_f32 fp32 fma chains, no range limit (about 3x slower on BASELINE cfg 2).
# 'auto'  'fast', but every call synchronises, and a call that left the range is re-run 'exact' before it returns.
_PRECISIONS = ('fast', 'exact', 'auto')
_precision = os.environ.get('STRIBOR_GEMM_PRECISION', 'fast')
if _precision not in _PRECISIONS:
    raise ValueError(f'STRIBOR_GEMM_PRECISION={_precision!r}; one of {_PRECISIONS}')


def set_gemm_precision(mode: str) -> str:
    """Select the conditioner-GEMM arithmetic for all later calls; returns the previous mode."""
    global _precision
    if mode not in _PRECISIONS:
        raise ValueError(f'gemm precision {mode!r}; one of {_PRECISIONS}')
    old, _precision = _precision, mode
    return old


def get_gemm_precision() -> str:
    return _precision


class GemmRangeError(OverflowError):
    """An operand of the fp16 x 3 conditioner GEMMs exceeded fp16's range (65504)."""


# ---- data-dependent error flags (reference: exceptions raised inside the torch ops) -----------------------------------
# One 32-bit word per (device, stream) in PINNED host memory: kernels OR SX_FLAG_* bits into it with system-scope atomics (the
# pinned allocation is mapped into the device's address space under the same pointer), the host reads it with a plain
# load -- no stream synchronisation, nothing that would break HIP-graph capture.  A kernel's flag becomes visible once
# that kernel has run, so `poll_errors()` at the start of the next call ON THE SAME STREAM reports the previous call's
# condition (work queued on another stream -- another flow, another thread -- neither sees nor clears it); `check_errors()`
# synchronises the device and raises whatever any of its streams left behind.
# Three reporting modes (STRIBOR_SYNC_ERRORS / set_sync_errors):
#   'grad' (default)  a NormalizingFlow call that builds an autograd graph (grad enabled, a parameter or the input requires it) of a
#                     flow that holds a rational-quadratic spline -- the reference op that asserts on its data
#                     (stribor/util/rational_quadratic_spline.py:175-178,223) -- synchronises its stream ONCE before it returns and
#                     raises what its kernels flagged: in a training loop the exception leaves the failing call like the reference's,
#                     at +2 .. 3 % of a cfg-3 training step (tools/experiments/sync_mode_cost.sh).  Flows without that op (the same
#                     rule would cost cfg 2 13 .. 24 % of a step), calls without a graph (inference, torch.no_grad()) and stream
#                     capture keep the deferred report of '0'.
#   '1' / True        every LAUNCH is followed by a stream synchronisation and the poll -- the debugging mode; not usable under
#                     HIP-graph capture.
#   '0' / False       never synchronise: the next call on the stream (or check_errors()) reports.
_flag_words = {}


def _parse_sync_mode(v) -> str:
    if isinstance(v, str):
        v = v.strip().lower()
        if v in ('grad', 'train'):
            return 'grad'
        return '0' if v in ('', '0', 'false', 'off', 'never') else '1'
    return '1' if v else '0'


_sync_mode = _parse_sync_mode(os.environ.get('STRIBOR_SYNC_ERRORS', 'grad'))
_sync_errors = _sync_mode == '1'          # per-launch synchronisation (read by call() / after_launch())


def set_sync_errors(on) -> str:
    """Select how data-dependent errors are reported: True / '1' inside the failing launch (synchronises after every launch), 'grad'
    at the end of every graph-building flow call (the default), False / '0' by the next call; returns the previous setting."""
    global _sync_errors, _sync_mode
    old = _sync_mode
    _sync_mode = _parse_sync_mode(on)
    _sync_errors = _sync_mode == '1'
    return old


def end_of_flow_call(x) -> None:
    """'grad' mode: called by NormalizingFlow's outermost public call when it built a graph -- wait for the stream and raise."""
    if _sync_mode == 'grad' and x.is_cuda and not torch.cuda.is_current_stream_capturing():
        with device_of(x):
            torch.cuda.current_stream().synchronize()
            poll_errors(device=x.device)


def _flag_entry(device):
    key = (torch.device(device).index or 0, stream())
    ent = _flag_words.get(key)
    if ent is None:
        if len(_flag_words) >= _MAX_FLAG_WORDS:
            _evict_flag_words()
        t = torch.zeros(1, dtype=torch.int32).pin_memory()
        ent = _flag_words[key] = (t, t.numpy(), t.data_ptr())
    return ent


_MAX_FLAG_WORDS = 64


def _evict_flag_words() -> None:
    """A program that keeps creating streams would grow the table by one pinned word per stream: beyond 64 entries, wait for the
    devices (no kernel may still hold a word's address), report what is pending and drop every word -- live streams get a new one at
    their next launch.  (A HIP graph captured earlier keeps the address of a dropped word: the words are retired, never freed, so a
    replay writes into memory nobody else owns; its flag is no longer polled -- INTEGRATION.md.)"""
    for idx in sorted({k[0] for k in _flag_words}):          # EVERY device that owns a word (torch.cuda.synchronize() alone waits for
        torch.cuda.synchronize(idx)                          # the current one: a pending flag elsewhere was read early and lost)
    pending = 0
    for ent in _flag_words.values():
        pending |= int(ent[1][0])
        ent[1][0] = 0
        _retired_words.append(ent[0])          # never handed back to the allocator: a captured graph may still write there
    _flag_words.clear()
    if pending:
        _raise_flags(pending)


_retired_words = []


def err_flag(device) -> int:
    """Pointer (host == device address) of the flag word of (`device`, its current stream)."""
    with device_of_index(torch.device(device).index):
        return _flag_entry(device)[2]


def _raise_flags(v: int) -> None:
    if v & FLAG_F16_RANGE:
        raise GemmRangeError(
            'an operand of the fp16 x 3 conditioner GEMMs (a weight, or a sample\'s flow state / hidden activation) '
            'exceeded 65504; the affected rows were returned as NaN.  Use stribor_amd.set_gemm_precision("exact") '
            '(fp32 MFMA, no range limit) or "auto" (re-runs such calls exactly).')
    if v & FLAG_RQS_NEG_DISCRIMINANT:
        raise AssertionError('rational_quadratic_spline: negative discriminant in the inverse pass')


def poll_errors(all_streams: bool = False, device=None) -> None:
    """Raise for any flag a COMPLETED kernel of the current stream of `device` (default: the current device) has set (no
    synchronisation); all_streams: of any stream of any device (what check_errors() does after synchronising)."""
    if all_streams:
        words = list(_flag_words.values())
    else:
        if not _flag_words:
            return
        if device is not None and torch.device(device).index is not None:
            with device_of_index(torch.device(device).index):          # the flag of x's device, not of whichever is current
                ent = _flag_words.get((torch.device(device).index, stream()))
        else:
            ent = _flag_words.get(((_cur_device() if _cur_device is not None else torch.cuda.current_device()), stream()))
        words = [] if ent is None else [ent]
    for _, view, _ in words:
        v = int(view[0])
        if v:
            view[0] = 0
            _raise_flags(v)


def after_launch() -> None:
    """STRIBOR_SYNC_ERRORS: wait for the launch just queued and raise its data-dependent error now."""
    if _sync_errors:
        torch.cuda.current_stream().synchronize()
        poll_errors()


def check_errors(device=None) -> None:
    """Synchronise and raise what the reference would have raised for data-dependent failures (any stream)."""
    if torch.cuda.is_available():
        torch.cuda.synchronize(device)
    poll_errors(all_streams=True)


def take_flag(device, bit: int) -> bool:
    """Clear `bit` of the flag word of (device, current stream) and tell whether it was set (callers synchronised already)."""
    with device_of_index(torch.device(device).index):
        ent = _flag_words.get((torch.device(device).index or 0, stream()))
    if ent is None:
        return False
    v = int(ent[1][0])
    if v & bit:
        ent[1][0] = v & ~bit
        return True
    return False


# ---- caller-owned scratch of the library (it allocates nothing itself) ------------------------------------------------
_work = {}
_scratch = {}
_POISON_SCRATCH = os.environ.get('STRIBOR_POISON_SCRATCH') == '1'


def work_counters(device) -> torch.Tensor:
    """The {ticket, done} pair of the fused kernel's dynamic chunk hand-out for (device, current stream)."""
    key = (torch.device(device).index or 0, stream())
    t = _work.get(key)
    if t is None:
        t = _work[key] = torch.zeros(2, dtype=torch.int32, device=device)
    return t


_redo = {}
_redo_off = threading.local()


def redo_list(device, n_rows: int) -> torch.Tensor:
    """The redo list of sx_flow_run2 for (device, current stream): sx_flow_redo_words(n_rows) zeroed 32-bit words; every call leaves
    it zeroed, so one buffer serves all launches of the stream (grown by replacement, like `scratch`)."""
    key = (torch.device(device).index or 0, stream())
    need = 2 + 2 * ((n_rows + 31) // 32)
    t = _redo.get(key)
    if t is None or t.numel() < need:
        t = _redo[key] = torch.zeros(max(need, 1 << 12), dtype=torch.int32, device=device)
    return t


class no_redo:
    """`with no_redo():` -- launches inside run without the exact redo pass (a graph-building flow call: a training step re-packs the
    weights every step, and the pass needs them packed a second time, for the exact kernels; such a call reports out-of-range samples
    through the flag word as before)."""

    def __enter__(self):
        _redo_off.n = getattr(_redo_off, 'n', 0) + 1

    def __exit__(self, *exc):
        _redo_off.n -= 1


def redo_allowed() -> bool:
    return getattr(_redo_off, 'n', 0) == 0 and os.environ.get('STRIBOR_NO_REDO', '0') in ('', '0')


def scratch(device, n_floats: int) -> torch.Tensor:
    """>= n_floats of fp32 scratch for (device, current stream); grown by replacement (the old block returns to torch's
    stream-ordered allocator, so launches already queued on this stream keep valid memory)."""
    key = (torch.device(device).index or 0, stream())
    t = _scratch.get(key)
    if t is None or t.numel() < n_floats:
        t = _scratch[key] = torch.empty(max(n_floats, 1 << 20), dtype=torch.float32, device=device)
    if _POISON_SCRATCH:          # debug: every hand-out starts as NaN, so a kernel reading what this op never wrote shows up
        t.fill_(float('nan'))
    return t


# ---- device guard -------------------------------------------------------------------------------------------------
# The library launches on the CURRENT device (hipGetDevice) and `stream()` is that device's current stream, so every
# launch runs under the device of the tensors it is given -- like torch's own ops (single-process multi-GPU use, or a
# caller that never called set_device).
_exchange = getattr(torch._C, '_cuda_exchangeDevice', None)


class device_of:
    """`with device_of(t):` makes t's device current for the block (no-op fast path when it already is)."""
    __slots__ = ('idx', 'prev')

    def __init__(self, t: torch.Tensor):
        self.idx = t.device.index if t.is_cuda else None

    def __enter__(self):
        self.prev = -1
        if self.idx is not None and _cur_device is not None and _cur_device() != self.idx:
            if _exchange is not None:
                self.prev = _exchange(self.idx)
            else:
                self.prev = torch.cuda.current_device()
                torch.cuda.set_device(self.idx)
        return self

    def __exit__(self, *exc):
        if self.prev >= 0:
            torch.cuda.set_device(self.prev)
        return False


class device_of_index(device_of):
    """`with device_of_index(i):` -- the same for a device index (None: leave the current device)."""
    __slots__ = ()

    def __init__(self, idx):
        self.idx = idx


def call(name: str, t: torch.Tensor, *args) -> None:
    """lib().<name>(*args, stream) on t's device and its current stream; raises on a non-zero status."""
    with device_of(t):
        rc = getattr(lib(), name)(*args, stream())
        if rc == 0 and _sync_errors:
            after_launch()
    if rc != 0:
        check(rc, name)


class PackTable:
    """sx_pack_linear_batch: a table of packs run as ONE launch.  `run(records, ...)` takes one tuple per pack,
    (W, b, out_dim, in_dim, row_idx, col_idx, m_tiles, k_tiles, row_scale, bias_scale, fold_ones, transpose, dst, bound_out) with device
    addresses (ints, 0 / None = NULL); the device copy of the table is rebuilt only when a record changes (parameters keep their
    storage across optimizer steps, so a training loop uploads it once)."""

    def __init__(self):
        self._key = None
        self._table = None          # device bytes of the records
        self._max_floats = 0

    def run(self, t: torch.Tensor, records, prec: int, flag_ptr) -> None:
        """on t's device and current stream; `t` only selects the device."""
        if not records:
            return
        key = tuple(records)
        if key != self._key or self._table is None or self._table.device != t.device:
            arr = (sx_pack_job * len(records))()
            mx = 0
            names = ('W', 'b', 'out_dim', 'in_dim', 'row_idx', 'col_idx', 'm_tiles', 'k_tiles', 'row_scale', 'bias_scale', 'fold_ones',
                     'transpose', 'dst', 'bound_out')
            pointers = {'W', 'b', 'row_idx', 'col_idx', 'row_scale', 'bias_scale', 'dst', 'bound_out'}
            for a, r in zip(arr, records):
                for name, v in zip(names, r):
                    setattr(a, name, (v or None) if name in pointers else v)
                mx = max(mx, packed_linear_floats(r[6], r[7]))
            host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
            # (a fresh device buffer per version: a launch already queued may still read the previous one)
            self._table = host.to(t.device)
            self._key, self._max_floats = key, mx
        call('sx_pack_linear_batch', t, self._table.data_ptr(), len(records), self._max_floats, prec, flag_ptr)


def packed_linear_floats(m_tiles: int, k_tiles: int) -> int:
    # pure arithmetic (mirrors sx_packed_linear_floats) so program layout can be planned without the library
    return m_tiles * k_tiles * 1024 + m_tiles * 32
