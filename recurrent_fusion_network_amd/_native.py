"""ctypes binding of librfn_hip.so (include/rfn.h).

The HIP library IS the product: there is no CPU or PyTorch-op fallback.  Importing this module
without the built library raises, and every call that returns a non-zero status raises.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# RFN_HIP_LIB: development hook (tools/ A/B builds of the same ABI, e.g. `make EXTRA=-D...` into another file); the product
# loads the library next to this file
LIB_PATH = os.environ.get('RFN_HIP_LIB') or os.path.join(_HERE, 'librfn_hip.so')

RFN_MAX_ENC = 8
RFN_GEMM_MAXSEG = 8
RFN_GEMM_MAXGROUP = 8
ABI_VERSION = 7
PATH_OPT_PERSIST_DEC_FWD, PATH_OPT_PERSIST_S2_FWD, PATH_OPT_PERSIST_DEC_BWD, PATH_OPT_PERSIST_S2_BWD = 1, 2, 4, 8   # rfn.h
PATH_OPT_PERSIST_ALL = 15
PATH_OPT_DEEP_CELLS = 16          # rfn.h RFN_PATH_OPT_DEEP_CELLS (A/B hook)
PATH_OPT_NO_SMALL_TILES = 32      # rfn.h RFN_PATH_OPT_NO_SMALL_TILES (A/B hook)
PATH_OPT_SHARED_SMALL_TILES = 64  # rfn.h RFN_PATH_OPT_SHARED_SMALL_TILES (A/B hook)
PATH_OPT_DEC_UNHOISTED = 128      # rfn.h RFN_PATH_OPT_DEC_UNHOISTED (A/B hook: the three-launch decoder cell of rounds 3-5)
CELL_VARIANT_DEEP = 256
GEMM_OPT_LDS_LEAN = 1
GEMM_OPT_NO_DMA = 2
GEMM_OPT_BF16X3 = 4
GEMM_OPT_BF16X3_ANY_SIZE = 8
GEMM_OPT_SPLITK_IN_KERNEL = 16
ADAM_MAXBUCKET = 16               # rfn.h RFN_ADAM_MAXBUCKET


class RfnError(RuntimeError):
    pass


class Dims(C.Structure):
    _fields_ = [('M', C.c_int32), ('R', C.c_int32), ('A', C.c_int32), ('E', C.c_int32), ('T1', C.c_int32),
                ('T2', C.c_int32), ('K', C.c_int32), ('V1', C.c_int32),
                ('L', C.c_int32 * RFN_MAX_ENC), ('D', C.c_int32 * RFN_MAX_ENC), ('F', C.c_int32 * RFN_MAX_ENC),
                ('review_maxout', C.c_int32), ('decoder_maxout', C.c_int32),
                ('drop_fusion', C.c_float), ('drop_reason', C.c_float), ('drop_lm', C.c_float),
                ('gemm_flags', C.c_uint32), ('path_flags', C.c_uint32), ('probe_events', C.c_void_p)]


class GemmSeg(C.Structure):
    _fields_ = [('A', C.c_void_p), ('B', C.c_void_p), ('bias', C.c_void_p), ('lda', C.c_int64),
                ('ldb', C.c_int64), ('K', C.c_int32), ('a_kfast', C.c_int32), ('b_kfast', C.c_int32),
                ('pad_', C.c_int32)]


class GemmLstm(C.Structure):
    _fields_ = [('c_prev', C.c_void_p), ('c_next', C.c_void_p), ('h_next', C.c_void_p), ('ldcp', C.c_int64), ('ldcn', C.c_int64),
                ('ldh', C.c_int64), ('gs_cprev', C.c_int64), ('gs_cnext', C.c_int64), ('gs_h', C.c_int64), ('drop_p', C.c_float),
                ('pad_', C.c_int32), ('seed', C.c_uint64), ('offset', C.c_uint64)]


class GemmProblem(C.Structure):
    _fields_ = [('C', C.c_void_p), ('ldc', C.c_int64), ('nseg', C.c_int32), ('pad_', C.c_int32),
                ('a_colsum', C.c_void_p), ('seg', GemmSeg * RFN_GEMM_MAXSEG)]


RFN_CELL_MAXOUT = 10
RFN_CELL_MAXSEG = 8
CELL_EPI_STORE, CELL_EPI_LSTM, CELL_EPI_LSTM_BWD = 0, 1, 2


class CellOut(C.Structure):
    _fields_ = [('C', C.c_void_p), ('ldc', C.c_int64), ('N', C.c_int32), ('accumulate', C.c_int32), ('nseg', C.c_int32),
                ('epilogue', C.c_int32), ('seg', GemmSeg * RFN_CELL_MAXSEG), ('c_prev', C.c_void_p), ('ldcp', C.c_int64),
                ('c_next', C.c_void_p), ('ldcn', C.c_int64), ('h_next', C.c_void_p), ('ldh', C.c_int64),
                ('drop_offset', C.c_uint64), ('gates', C.c_void_p), ('ldg', C.c_int64), ('dh_ext', C.c_void_p),
                ('lddh', C.c_int64), ('dc_next', C.c_void_p), ('lddcn', C.c_int64), ('dc_prev', C.c_void_p),
                ('lddcp', C.c_int64), ('acc_slabs', C.c_void_p), ('acc_parts', C.c_int32), ('acc_stride', C.c_int64)]


def _load():
    if not os.path.exists(LIB_PATH):
        raise RfnError(
            'librfn_hip.so is missing (%s). Build it with `python -c "import __graft_entry__ as g; g.build()"` '
            'or `make -C recurrent_fusion_network_amd/csrc`. There is no fallback path.' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    if lib.rfn_abi_version() != ABI_VERSION:
        raise RfnError('librfn_hip.so ABI %d != binding ABI %d: rebuild' % (lib.rfn_abi_version(), ABI_VERSION))
    lib.rfn_error_string.restype = C.c_char_p
    P, I, L, F, SZ, U64 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t, C.c_uint64
    DP = C.POINTER(Dims)
    sig = {
        'rfn_param_count': (C.c_int, [DP]),
        'rfn_param_name': (C.c_int, [DP, I, C.c_char_p, SZ]),
        'rfn_param_shape': (C.c_int, [DP, I, C.POINTER(L), C.POINTER(L)]),
        'rfn_gemm_f32': (C.c_int, [I, I, I, C.POINTER(GemmProblem), I, P]),
        'rfn_gemm_f32_ws': (C.c_int, [I, I, I, C.POINTER(GemmProblem), I, P, SZ, P]),
        'rfn_gemm_f32_opt': (C.c_int, [I, I, I, C.POINTER(GemmProblem), I, P, SZ, C.c_uint, P]),
        'rfn_gemm_f32_tk': (C.c_int, [I, I, I, C.POINTER(GemmProblem), I, P, SZ, C.c_uint, P, I, P]),
        'rfn_gemm_f32_lstm': (C.c_int, [I, I, I, C.POINTER(GemmProblem), P, SZ, C.c_uint, C.POINTER(GemmLstm), P]),
        'rfn_x3_image_bytes': (SZ, [I, I]),
        'rfn_x3_split': (C.c_int, [P, I, L, I, I, I, P, P]),
        'rfn_x3_gemm': (C.c_int, [I, I, I, P, P, I, I, P, P, L, I, I, P, P]),
        'rfn_x3_split_ks': (C.c_int, [P, I, L, I, I, P, P]),
        'rfn_x3_gemm_ks': (C.c_int, [I, I, I, P, P, I, I, P, P, L, I, I, P, P]),
        'rfn_attn_bwd_grouped_ks': (C.c_int, [I, P, L, L, P, P, P, P, L, L, P, L, I, I, I, I, P, I, I, P, P, P]),
        'rfn_x3_part_floats': (SZ, [I, I, I]),
        'rfn_x3_splitk_for': (C.c_int, [I, I, I]),
        'rfn_colsum_f32': (C.c_int, [P, L, I, I, P, I, P]),
        'rfn_colsum_grouped_f32': (C.c_int, [P, L, L, I, I, P, I, P]),
        'rfn_colsum_grouped2_f32': (C.c_int, [P, L, L, I, I, P, P, I, P]),
        'rfn_fill_small_f32': (C.c_int, [P, I, I, F, P]),
        'rfn_copy_small_f32': (C.c_int, [P, P, I, I, P]),
        'rfn_attn_scores_fwd': (C.c_int, [P, L, L, P, P, P, I, I, I, P, P]),
        'rfn_attn_context_fwd': (C.c_int, [P, L, L, P, I, I, I, P, L, P]),
        'rfn_attn_context_bwd_dalpha': (C.c_int, [P, L, L, P, L, I, I, I, P, P]),
        'rfn_attn_context_bwd_dseq': (C.c_int, [P, P, L, I, I, I, P, L, L, P]),
        'rfn_attn_scores_bwd': (C.c_int, [P, L, L, P, P, P, P, I, I, I, P, L, L, I, P, P, P]),
        'rfn_attn_fwd': (C.c_int, [P, L, L, P, P, P, P, L, L, I, I, I, I, P, P, P, L, P]),
        'rfn_attn_bwd': (C.c_int, [P, L, L, P, P, P, P, L, L, P, L, I, I, I, I, P, L, L, I, P, P, P]),
        'rfn_attn_fwd_grouped': (C.c_int, [I, P, L, L, P, P, P, P, L, L, I, I, I, I, P, P, P, L, P]),
        'rfn_attn_bwd_grouped': (C.c_int, [I, P, L, L, P, P, P, P, L, L, P, L, I, I, I, I, P, L, L, I, P, P, P]),
        'rfn_attn_fwd_het': (C.c_int, [I, P, P, P, P, P, I, P, I, P, P, P, P, P]),
        'rfn_attn_bwd_het': (C.c_int, [I, P, P, P, P, P, P, I, P, I, P, P, I, P, P, P]),
        'rfn_attn_small_fwd': (C.c_int, [I, P, L, L, P, P, P, P, L, L, I, I, I, I, P, P, L, P]),
        'rfn_attn_small_bwd': (C.c_int, [I, P, L, L, P, P, P, P, L, L, P, L, I, I, I, I, P, L, L, I, P, P, P, P]),
        'rfn_dropout_mask': (C.c_int, [U64, U64, L, F, P, P]),
        'rfn_cell_gemm_supported': (C.c_int, [I, I, C.POINTER(CellOut), I]),
        'rfn_cell_gemm': (C.c_int, [I, I, C.POINTER(CellOut), I, F, U64, I, P]),
        'rfn_lstm_fwd': (C.c_int, [P, L, P, L, P, L, P, L, I, I, I, F, U64, U64, P]),
        'rfn_lstm_bwd': (C.c_int, [P, L, P, L, P, L, P, L, P, L, P, L, I, I, I, F, U64, U64, P]),
        'rfn_lstm_fwd_grouped': (C.c_int, [P, L, P, L, P, L, P, L, I, I, I, F, U64, U64, I, L, L, L, L, P]),
        'rfn_lstm_bwd_grouped': (C.c_int, [P, L, P, L, P, L, P, L, P, L, P, L, I, I, I, F, U64, U64, I, L, L, L, L, P]),
        'rfn_embed_fwd': (C.c_int, [P, I, L, P, I, L, L, I, P, L, P]),
        'rfn_embed_bwd': (C.c_int, [P, L, P, I, L, L, I, I, L, P, P]),
        'rfn_log_softmax_fwd': (C.c_int, [P, L, I, I, I, L, L, P, P]),
        'rfn_log_softmax_bwd': (C.c_int, [P, P, I, I, I, L, L, P, L, P]),
        'rfn_max_over_steps_fwd': (C.c_int, [P, I, I, I, P, P, P]),
        'rfn_max_over_steps_bwd': (C.c_int, [P, P, I, I, I, P, P]),
        'rfn_max_over_steps_fwd_grouped': (C.c_int, [P, I, I, I, P, P, I, P]),
        'rfn_max_over_steps_bwd_grouped': (C.c_int, [P, P, I, I, I, P, I, P]),
        'rfn_axpby_2d': (C.c_int, [F, P, L, F, P, L, I, I, P]),
        'rfn_div_2d': (C.c_int, [P, L, I, I, F, P]),
        'rfn_mean_over_groups': (C.c_int, [I, P, L, L, I, P, L, I, I, P]),
        'rfn_bcast_to_groups': (C.c_int, [I, F, P, L, P, P, L, L, I, I, I, P]),
        'rfn_xe_logits_fwd': (C.c_int, [P, L, I, I, I, P, L, P, L, F, P, P, P, I, P]),
        'rfn_xe_logits_bwd': (C.c_int, [P, L, I, I, I, P, L, P, L, F, P, F, P, P]),
        'rfn_decoder_logits': (P, [DP, I, I, I, P]),
        'rfn_xe_loss': (C.c_int, [P, I, I, I, P, L, P, L, F, F, P, P, I, P, P]),
        'rfn_multilabel_margin': (C.c_int, [P, I, I, P, F, F, P, P, I, P, P]),
        'rfn_xe_loss_ex': (C.c_int, [P, I, I, I, P, L, P, L, F, F, P, P, P, I, P, P]),
        'rfn_multilabel_margin_grouped': (C.c_int, [I, P, I, I, P, F, F, P, P, P, I, P, P]),
        'rfn_rl_loss': (C.c_int, [P, L, P, L, P, L, P, L, L, I, I, I, F, P, L, I, F, P, P, I, P, L, P, L, L, P]),
        'rfn_rl_loss_ex': (C.c_int, [P, L, P, L, P, L, P, L, L, I, I, I, I, F, P, L, I, F, P, P, P, I, P, L, P, L, L, P]),
        'rfn_adam_step': (C.c_int, [P, P, P, P, L, F, F, F, F, F, F, F, I, P]),
        'rfn_adam_step_multi': (C.c_int, [I, P, P, P, P, P, F, F, F, F, F, F, F, I, P]),
        'rfn_adam_step_multi_coef': (C.c_int, [I, P, P, P, P, P, P, F, F, F, F, F, F, P]),
        'rfn_greedy_pick': (C.c_int, [P, L, I, I, I, P, P, L, P, L, P, P, P]),
        'rfn_multinomial_pick': (C.c_int, [P, L, I, I, F, P, P, F, P, L, P]),
        'rfn_beam_step': (C.c_int, [P, L, I, I, I, I, I, I, P, P, P, P, P, P, P, P, P, P, P]),
        'rfn_gather_rows': (C.c_int, [P, P, P, I, I, P]),
        'rfn_beam_step_topk': (C.c_int, [P, P, I, I, I, I, I, I, P, P, P, P, P, P, P, P, P, P, P]),
        'rfn_log_softmax_topk': (C.c_int, [P, L, I, I, I, P, P, P]),
        'rfn_prefix_ws_bytes': (SZ, [DP, I, I]),
        'rfn_prefix_fwd': (C.c_int, [DP, I, P, P, P, P, P, P, P, P, SZ, I, U64, P]),
        'rfn_prefix_fwd_from_state': (C.c_int, [DP, I, P, P, P, P, P, P, P, P, P, SZ, P]),
        'rfn_prefix_bwd': (C.c_int, [DP, I, P, P, P, P, P, P, P, P, P, SZ, U64, I, P]),
        'rfn_prefix_bwd_wgrad': (C.c_int, [DP, I, P, P, P, SZ, I, I, P]),
        'rfn_decoder_ws_bytes': (SZ, [DP, I, I, I]),
        'rfn_decoder_fwd': (C.c_int, [DP, I, I, P, P, P, P, P, L, P, P, SZ, I, U64, P]),
        'rfn_decoder_fwd_begin': (C.c_int, [DP, I, I, P, P, P, P, P, SZ, I, P]),
        'rfn_decoder_fwd_step': (C.c_int, [DP, I, I, I, P, P, P, L, P, P, SZ, I, U64, P]),
        'rfn_decoder_bwd': (C.c_int, [DP, I, I, P, P, P, P, P, L, P, P, P, P, P, P, P, SZ, U64, P]),
        'rfn_pick_record': (C.c_int, [P, L, I, I, I, P, P, P, L, P, L, P, P, P]),
        'rfn_decoder_loop': (C.c_int, [DP, I, I, P, P, P, P, P, I, F, P, P, L, L, P, L, P, L, P, P, P, SZ, U64, P]),
        'rfn_decoder_fwd_sampled': (C.c_int, [DP, I, I, P, P, P, P, P, L, F, F, P, P, P, P, SZ, I, U64, P]),
        'rfn_beam_loop': (C.c_int, [DP, I, I, I] + [P] * 18 + [I, P, SZ, U64, P]),
        'rfn_decoder_step_ws_bytes': (SZ, [DP, I]),
        'rfn_decoder_prepare': (C.c_int, [DP, I, P, P, P, P]),
        'rfn_decoder_cproj_floats': (SZ, [DP, I]),
        'rfn_dec_cell_fwd': (C.c_int, [P, L, L, P, P, P, P, L, L, P, P, L, P, L, P, L, P, L, P, I, I, I, I, I, I, C.c_float, U64, U64, P]),
        'rfn_dec_attn_bwd': (C.c_int, [P, L, L, P, P, P, P, L, L, P, L, I, I, I, I, P, L, L, I, P, P, P]),
        'rfn_dec_du': (C.c_int, [P, P, I, I, I, I, P, L, L, P]),
        'rfn_decoder_step': (C.c_int, [DP, I, P, P, P, P, P, P, P, P, L, P, SZ, U64, I, P]),
        'rfn_decoder_step_embedded': (C.c_int, [DP, I, P, P, P, P, L, P, P, P, P, L, P, SZ, U64, I, P]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib, sorted(sig) + ['rfn_abi_version', 'rfn_error_string']


lib, EXPORTS = _load()


def check(rc: int, what: str = '') -> None:
    if rc != 0:
        raise RfnError('%s failed: %s (code %d)' % (what or 'librfn_hip call', lib.rfn_error_string(rc).decode(), rc))


def stream_ptr() -> int:
    """The current PyTorch HIP stream as a raw hipStream_t."""
    return torch.cuda.current_stream().cuda_stream


def ptr(t) -> int:
    return 0 if t is None else t.data_ptr()


def require_cuda_f32(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise RfnError('%s must live on the GPU: the HIP path has no CPU fallback' % name)
    if t.dtype != torch.float32:
        raise RfnError('%s must be float32, got %s' % (name, t.dtype))
    return t.contiguous()


def ptr_array(tensors):
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else t.data_ptr()
    return arr


def make_dims(M, R, A, E, T1, T2, K, V1, L, D, Fc, review_maxout=0, decoder_maxout=0, drop_fusion=0.0,
              drop_reason=0.0, drop_lm=0.0, gemm_flags=0, path_flags=0) -> Dims:
    if M > RFN_MAX_ENC:
        raise RfnError('at most %d encoders are supported' % RFN_MAX_ENC)
    d = Dims()
    d.M, d.R, d.A, d.E, d.T1, d.T2, d.K, d.V1 = M, R, A, E, T1, T2, K, V1
    for i in range(M):
        d.L[i], d.D[i], d.F[i] = L[i], D[i], Fc[i]
    d.review_maxout, d.decoder_maxout = int(review_maxout), int(decoder_maxout)
    d.drop_fusion, d.drop_reason, d.drop_lm = drop_fusion, drop_reason, drop_lm
    d.gemm_flags = int(gemm_flags)
    d.path_flags = int(path_flags)
    d.probe_events = None
    return d


def param_names(d: Dims):
    n = lib.rfn_param_count(C.byref(d))
    if n < 0:
        check(n, 'rfn_param_count')
    buf = C.create_string_buffer(160)
    out = []
    for i in range(n):
        check(lib.rfn_param_name(C.byref(d), i, buf, 160), 'rfn_param_name')
        out.append(buf.value.decode())
    return out


def param_shape(d: Dims, idx: int):
    r, c = C.c_int64(), C.c_int64()
    check(lib.rfn_param_shape(C.byref(d), idx, C.byref(r), C.byref(c)), 'rfn_param_shape')
    return r.value, c.value


# ---- thin helpers over the primitive operators (used by the model shell and by the tests) ---------
def gemm(M, N, problems, accumulate=False, ws=None, flags=0, tickets=None):
    """problems: list of (C, ldc, [(A, lda, a_kfast, B, ldb, b_kfast, K, bias), ...][, a_colsum]).
    ws: optional uint8 scratch tensor enabling split-K for skinny problems; flags: GEMM_OPT_* bits; tickets: optional
    zeroed int32 tensor -> split-K is finished inside the launch (rfn_gemm_f32_tk)."""
    arr = (GemmProblem * len(problems))()
    for g, prob in enumerate(problems):
        Ct, ldc, segs = prob[:3]
        arr[g].C, arr[g].ldc, arr[g].nseg = Ct.data_ptr(), ldc, len(segs)
        arr[g].a_colsum = ptr(prob[3]) if len(prob) > 3 else None
        for s, (A, lda, ak, B, ldb, bk, K, bias) in enumerate(segs):
            sg = arr[g].seg[s]
            sg.A, sg.lda, sg.a_kfast = A.data_ptr(), lda, int(ak)
            sg.B, sg.ldb, sg.b_kfast = B.data_ptr(), ldb, int(bk)
            sg.K, sg.bias = K, ptr(bias)
    check(lib.rfn_gemm_f32_tk(M, N, len(problems), arr, int(accumulate), ptr(ws), 0 if ws is None else ws.numel(),
                              int(flags), ptr(tickets), 0 if tickets is None else tickets.numel(), stream_ptr()),
          'rfn_gemm_f32_tk')


def cell_gemm_args(outs):
    """The rfn_cell_out array of `outs` (see cell_gemm); keep the tensors alive while it is used."""
    return _cell_outs(outs)


def cell_gemm(M, outs, R=0, drop_p=0.0, seed=0, variant=0):
    """Row-panel GEMM of the recurrences (rfn_cell_gemm).  outs: list of dicts with C, ldc, N, accumulate, segs =
    [(A, lda, B, ldb, b_kfast, K, bias), ...] and, for the LSTM gate epilogue, lstm = (c_prev, ldcp, c_next, ldcn, h_next,
    ldh, drop_offset); for the gate-gradient epilogue lstm_bwd = (gates, ldg, c_prev, ldcp, c_next, ldcn, dh_ext, lddh,
    dc_next, lddcn, dc_prev, lddcp, drop_offset) and C may be None; acc = (slabs, parts, stride): partial slabs added to C."""
    check(lib.rfn_cell_gemm(M, len(outs), _cell_outs(outs), R, drop_p, seed, variant, stream_ptr()), 'rfn_cell_gemm')


def _cell_outs(outs):
    arr = (CellOut * len(outs))()
    for o, spec in enumerate(outs):
        t = arr[o]
        t.C, t.ldc, t.N, t.accumulate = ptr(spec['C']), spec['ldc'], spec['N'], int(spec.get('accumulate', 0))
        t.nseg = len(spec['segs'])
        t.epilogue = CELL_EPI_LSTM if 'lstm' in spec else (CELL_EPI_LSTM_BWD if 'lstm_bwd' in spec else CELL_EPI_STORE)
        for s, (A, lda, B, ldb, bk, K, bias) in enumerate(spec['segs']):
            sg = t.seg[s]
            sg.A, sg.lda, sg.a_kfast = A.data_ptr(), lda, 1
            sg.B, sg.ldb, sg.b_kfast = B.data_ptr(), ldb, int(bk)
            sg.K, sg.bias = K, ptr(bias)
        if 'acc' in spec:      # accumulate: C + `parts` slabs of C's shape at slabs + p * stride (rfn.h, rfn_cell_out)
            slabs, parts, stride = spec['acc']
            t.acc_slabs, t.acc_parts, t.acc_stride = slabs.data_ptr(), int(parts), int(stride)
        if 'lstm' in spec:
            cp, ldcp, cn, ldcn, hn, ldh, off = spec['lstm']
            t.c_prev, t.ldcp, t.c_next, t.ldcn, t.h_next, t.ldh, t.drop_offset = (cp.data_ptr(), ldcp, cn.data_ptr(), ldcn,
                                                                                 hn.data_ptr(), ldh, off)
        if 'lstm_bwd' in spec:
            g, ldg, cp, ldcp, cn, ldcn, dhe, lddh, dcn, lddcn, dcp, lddcp, off = spec['lstm_bwd']
            t.gates, t.ldg, t.c_prev, t.ldcp, t.c_next, t.ldcn = g.data_ptr(), ldg, cp.data_ptr(), ldcp, cn.data_ptr(), ldcn
            t.dh_ext, t.lddh, t.dc_next, t.lddcn, t.dc_prev, t.lddcp, t.drop_offset = (ptr(dhe), lddh, ptr(dcn), lddcn,
                                                                                       dcp.data_ptr(), lddcp, off)
    return arr


def x3_image(srcs, rows, K, k_fast=True, ld=None) -> torch.Tensor:
    """bf16 plane image (uint8 tensor) of the logical operand whose row block g is the f32 matrix srcs[g]
    (rows x K; k_fast: element (row, k) at [row * ld + k], else at [k * ld + row])."""
    srcs = [require_cuda_f32(t, 'src') for t in srcs]
    if ld is None:
        ld = K if k_fast else rows
    img = torch.empty(lib.rfn_x3_image_bytes(len(srcs) * rows, K), dtype=torch.uint8, device=srcs[0].device)
    check(lib.rfn_x3_split(ptr_array(srcs), len(srcs), ld, rows, K, int(k_fast), img.data_ptr(), stream_ptr()), 'rfn_x3_split')
    return img


def x3_image_ks(srcs, K, cols, ld=None) -> torch.Tensor:
    """k-slow bf16 plane image (uint8 tensor) of Y[K][len(srcs) * cols] whose column block g is srcs[g][k * ld + c]."""
    srcs = [require_cuda_f32(t, 'src') for t in srcs]
    img = torch.empty(lib.rfn_x3_image_bytes(len(srcs) * cols, K), dtype=torch.uint8, device=srcs[0].device)
    check(lib.rfn_x3_split_ks(ptr_array(srcs), len(srcs), cols if ld is None else ld, K, cols, img.data_ptr(), stream_ptr()),
          'rfn_x3_split_ks')
    return img


def x3_gemm(M, N, K, img_a, img_b, outs, gm=None, gn=None, ldc=None, bias=None, accumulate=False, splitk=1, k_slow=False):
    """outs: list of f32 output tensors, group (i, j) = outs[i * ceil(N / gn) + j] (gm x gn each); bias: same shape list or None.
    k_slow: both images are k-slow images (x3_image_ks)."""
    gm, gn = gm or M, gn or N
    ldc = ldc or gn
    part = None
    if splitk > 1:
        part = torch.empty(lib.rfn_x3_part_floats(M, N, splitk), dtype=torch.float32, device=img_a.device)
    fn = lib.rfn_x3_gemm_ks if k_slow else lib.rfn_x3_gemm
    check(fn(M, N, K, img_a.data_ptr(), img_b.data_ptr(), gm, gn, ptr_array(outs),
             None if bias is None else ptr_array(bias), ldc, int(accumulate), splitk, ptr(part), stream_ptr()), 'rfn_x3_gemm')


def linear(x: torch.Tensor, weight: torch.Tensor, bias=None) -> torch.Tensor:
    """y = x W^T + b through the MFMA GEMM (no autograd)."""
    x = require_cuda_f32(x, 'x')
    weight = require_cuda_f32(weight, 'weight')
    rows, K = x.shape
    N = weight.shape[0]
    y = torch.empty(rows, N, device=x.device, dtype=torch.float32)
    gemm(rows, N, [(y, N, [(x, K, 1, weight, K, 1, K, bias)])])
    return y
