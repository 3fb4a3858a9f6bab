"""ctypes binding of ``libpgca_hip.so`` (C ABI: ``include/pgca_hip.h``).

There is NO CPU fallback: if the library is missing or a call fails, a
``RuntimeError`` is raised.  Wrappers take torch tensors that live on the GPU,
pass ``data_ptr()`` + sizes and launch on torch's current HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# PGCA_LIB: a diagnostic build of the same ABI (python -m pgca_amd.build --variant=timing -DPGCA_GEMM_TIMING); the product
# library is the in-tree libpgca_hip.so
LIB_PATH = os.environ.get("PGCA_LIB") or os.path.join(_HERE, "libpgca_hip.so")

NT, NN, TN = 0, 1, 2
EPI_NONE, EPI_GELU_NEW, EPI_QUICK_GELU, EPI_RELU, EPI_TANH = 0, 1, 2, 3, 4
EPI_DGELU_NEW, EPI_DRELU, EPI_DTANH, EPI_ROWSTATS, EPI_DLOGITS, EPI_DQUICK_GELU = 5, 6, 7, 8, 9, 10
EPI_GELU_NEW_D, EPI_MUL_AUX = 11, 12

_vp, _i32, _i64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float


class GemmArgs(C.Structure):
    _fields_ = [
        ("A", _vp), ("B", _vp),
        ("M", _i32), ("N", _i32), ("K", _i32),
        ("lda", _i32), ("ldb", _i32),
        ("layout", _i32), ("epilogue", _i32),
        ("alpha", _f32),
        ("bias", _vp),
        ("out_bf16", _vp), ("ld_out_bf16", _i32),
        ("out_f32", _vp), ("ld_out_f32", _i32),
        ("accumulate", _i32),
        ("residual", _vp), ("ld_res", _i32),
        ("aux_out", _vp), ("aux_in", _vp), ("ld_aux", _i32),
        ("targets", _vp), ("stat_max", _vp), ("stat_sum", _vp), ("stat_ld", _i32),
        ("target_val", _vp), ("row_lse", _vp), ("row_scale", _vp),
        ("out_cols", _i32),
        ("drop_seed", C.c_uint32), ("drop_threshold", C.c_uint32), ("drop_scale", _f32),
        ("colsum_part", _vp), ("ld_colsum", _i32),
        ("drop_rows", _vp),
    ]


class SkinnyArgs(C.Structure):
    _fields_ = [
        ("x", _vp), ("W", _vp),
        ("M", _i32), ("N", _i32), ("K", _i32), ("lda", _i32), ("ldw", _i32),
        ("scratch", _vp), ("bias", _vp), ("act", _i32),
        ("residual", _vp), ("ld_res", _i32),
        ("out_f32", _vp), ("ld_out_f32", _i32),
        ("out_bf16", _vp), ("ld_out_bf16", _i32),
        ("ln_gamma", _vp), ("ln_beta", _vp), ("ln_eps", _f32),
        ("ln_out_bf16", _vp), ("ld_ln", _i32),
    ]


SKINNY_MAX_M = 64  # include/pgca_hip.h PGCA_SKINNY_MAX_M

# name -> argtypes (return type is always int status unless noted)
_SIGS = {
    "pgca_gemm_bf16": [C.POINTER(GemmArgs), _vp],
    "pgca_gemm_plan": [C.POINTER(GemmArgs)],
    "pgca_set_option": [C.c_char_p, _i32],
    "pgca_gemm_bf16_grouped": [C.POINTER(GemmArgs), _i32, _vp],
    "pgca_gemm_skinny": [C.POINTER(SkinnyArgs), _vp],
    "pgca_rowstats_combine": [_vp, _vp, _i32, _i32, _vp, _i32, _vp, _vp, _vp],
    "pgca_layernorm_fwd": [_vp, _vp, _i32, _i32, _vp, _vp, _f32, _vp, _vp, _vp, _vp, _vp],
    "pgca_layernorm_bwd_blocks": [_i32],
    "pgca_layernorm_bwd": [_vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "pgca_colsum_finish": [_vp, _i32, _i32, _vp, _i32, _vp],
    "pgca_colsum_finish4": [_vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _vp],
    "pgca_colsum_blocks": [_i32],
    "pgca_colsum": [_vp, _vp, _i32, _i32, _i32, _vp, _vp],
    "pgca_attention_fwd": [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, C.c_uint32, C.c_uint32, _f32, _vp, _vp],
    "pgca_attention_bwd": [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, C.c_uint32, C.c_uint32, _f32, _vp, _vp],
    "pgca_embed_fwd": [_vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _f32, _vp, _vp, _vp, _i32, _vp, _i32, _vp, _vp,
                       _vp, _i32, _vp],
    "pgca_embed_bwd": [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp,
                       _i32, _vp, _vp, _vp, _vp],
    "pgca_embed_bwd_blocks": [_i32, _i32],
    "pgca_patchify": [_vp, _i32, _i32, _i32, _i32, _vp, _vp],
    "pgca_image_preprocess": [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _vp, _i32, _f32, _f32, _f32, _f32, _f32, _f32,
                              _vp, _vp, _vp, _vp],
    "pgca_image_train_transform": [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _i32, _f32, _f32, _f32,
                                   _f32, _f32, _f32, _vp, _vp, _vp, _vp, _vp],
    "pgca_vit_assemble": [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp],
    "pgca_vit_assemble_bwd": [_vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp],
    "pgca_seq_reduce": [_vp, _vp, _i32, _i32, _vp, _i32, _vp, _vp],
    "pgca_logits_logprob": [_vp, _i32, _i32, _vp, _vp, _i32, _vp, _vp],
    "pgca_logits_logprob_bwd": [_vp, _i32, _i32, _vp, _vp, _vp, _i32, _vp, _vp],
    "pgca_dpo_loss": [_vp, _vp, _vp, _vp, _i32, _f32, _f32, _vp, _vp, _vp, _vp, _vp],
    "pgca_row_scale": [_vp, _vp, _vp, _i32, _i32, _vp, _vp],
    "pgca_seq_batch_prepare": [_vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "pgca_seq_pack_prepare": [_vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "pgca_masked_mean_fwd": [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp],
    "pgca_masked_mean_bwd": [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp],
    "pgca_l2norm_fwd": [_vp, _i32, _i32, _vp, _vp, _vp],
    "pgca_l2norm_bwd": [_vp, _vp, _vp, _i32, _i32, _vp, _vp],
    "pgca_ntxent_loss": [_vp, _vp, _vp, _i32, _i32, _vp, _vp],
    "pgca_sqnorm_blocks": [_i64],
    "pgca_sqnorm": [_vp, _i64, _vp, _vp],
    "pgca_step_control": [_vp, _i32, _f32, _f32, _i32, _i32, _i32, _f32, _f32, _f32, _vp, _vp, _vp],
    "pgca_clip_coef": [_vp, _i32, _f32, _vp, _vp],
    "pgca_scale_dev": [_vp, _i64, _vp, _vp],
    "pgca_adamw": [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _f32, _f32, _f32, _f32, _f32, _vp],
    "pgca_cast_bf16": [_vp, _vp, _i64, _vp],
    "pgca_cast_f32": [_vp, _vp, _i64, _vp],
    "pgca_split_bf16": [_vp, _i32, _i32, _i32, _i32, _vp, _vp],
    "pgca_axpy": [_vp, _f32, _vp, _i64, _i32, _vp],
    "pgca_gather_rows_bf16": [_vp, _vp, _i32, _i32, _vp, _vp],
}
EXPORTS = ["pgca_version", "pgca_last_error", "pgca_sizeof_gemm_args", "pgca_sizeof_skinny_args",
           "pgca_gemm_skinny_workspace"] + list(_SIGS)
ABI_VERSION = 304  # include/pgca_hip.h PGCA_ABI_VERSION

_lib = None


def load() -> C.CDLL:
    """Load the HIP library (after torch, so libamdhip64.so.7 resolves to the loaded runtime)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the MI355X path has no fallback. Build it with "
            "`python -m pgca_amd.build` (hipcc --offload-arch=gfx950).")
    lib = C.CDLL(LIB_PATH)
    lib.pgca_version.restype = C.c_int
    lib.pgca_last_error.restype = C.c_char_p
    if lib.pgca_version() != ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH} has ABI version {lib.pgca_version()}, this binding expects {ABI_VERSION}: "
                           "rebuild it with `python -m pgca_amd.build --force`")
    lib.pgca_sizeof_gemm_args.restype = C.c_int
    if lib.pgca_sizeof_gemm_args() != C.sizeof(GemmArgs):
        raise RuntimeError(f"pgca_gemm_args is {lib.pgca_sizeof_gemm_args()} bytes in {LIB_PATH} but "
                           f"{C.sizeof(GemmArgs)} in the binding: stale library, rebuild it")
    lib.pgca_sizeof_skinny_args.restype = C.c_int
    if lib.pgca_sizeof_skinny_args() != C.sizeof(SkinnyArgs):
        raise RuntimeError(f"pgca_skinny_args is {lib.pgca_sizeof_skinny_args()} bytes in {LIB_PATH} but "
                           f"{C.sizeof(SkinnyArgs)} in the binding: stale library, rebuild it")
    lib.pgca_gemm_skinny_workspace.restype = C.c_int64
    lib.pgca_gemm_skinny_workspace.argtypes = [_i32, _i32, _i32]
    for name, sig in _SIGS.items():
        fn = getattr(lib, name)
        fn.argtypes = sig
        fn.restype = C.c_int
    _lib = lib
    return lib


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what} failed ({rc}): {load().pgca_last_error().decode()}")


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


# --------------------------------------------------------------------------- GEMM
def gemm(A: torch.Tensor, B: torch.Tensor, M: int, N: int, K: int, layout: int, *, lda: int = None, ldb: int = None,
         epilogue: int = EPI_NONE, alpha: float = 1.0, bias: torch.Tensor = None,
         out_bf16: torch.Tensor = None, ld_out_bf16: int = None, out_f32: torch.Tensor = None, ld_out_f32: int = None,
         accumulate: bool = False, residual: torch.Tensor = None, ld_res: int = None,
         aux_out: torch.Tensor = None, aux_in: torch.Tensor = None, ld_aux: int = None,
         targets: torch.Tensor = None, stat_max: torch.Tensor = None, stat_sum: torch.Tensor = None, stat_ld: int = 0,
         target_val: torch.Tensor = None, row_lse: torch.Tensor = None, row_scale: torch.Tensor = None,
         out_cols: int = 0, drop=None, colsum_part: torch.Tensor = None, drop_rows: torch.Tensor = None) -> None:
    a = GemmArgs()
    a.A, a.B = A.data_ptr(), B.data_ptr()
    a.M, a.N, a.K = M, N, K
    a.lda = lda if lda is not None else (K if layout != TN else M)
    a.ldb = ldb if ldb is not None else (K if layout == NT else N)
    a.layout, a.epilogue, a.alpha = layout, epilogue, alpha
    a.bias = _p(bias)
    a.out_bf16, a.ld_out_bf16 = _p(out_bf16), (ld_out_bf16 if ld_out_bf16 is not None else N)
    a.out_f32, a.ld_out_f32 = _p(out_f32), (ld_out_f32 if ld_out_f32 is not None else N)
    a.accumulate = 1 if accumulate else 0
    a.residual, a.ld_res = _p(residual), (ld_res if ld_res is not None else N)
    a.aux_out, a.aux_in, a.ld_aux = _p(aux_out), _p(aux_in), (ld_aux if ld_aux is not None else N)
    a.targets, a.stat_max, a.stat_sum, a.stat_ld = _p(targets), _p(stat_max), _p(stat_sum), stat_ld
    a.target_val, a.row_lse, a.row_scale = _p(target_val), _p(row_lse), _p(row_scale)
    a.out_cols = out_cols
    if drop is not None:
        a.drop_seed, a.drop_threshold, a.drop_scale = drop
        a.drop_rows = _p(drop_rows)
    if colsum_part is not None:
        a.colsum_part, a.ld_colsum = colsum_part.data_ptr(), colsum_part.shape[1]
    probe = gemm_probe
    if probe is not None and probe.want(layout, epilogue, load().pgca_gemm_plan(C.byref(a))):
        # HIP events on the launch stream bracket this one kernel (bench.py roofline measurement)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _check(load().pgca_gemm_bf16(C.byref(a), _stream()), "pgca_gemm_bf16")
        e1.record()
        probe.add(e0, e1, 2.0 * M * N * K)
        return
    _check(load().pgca_gemm_bf16(C.byref(a), _stream()), "pgca_gemm_bf16")


def gemm_wgrad_group(problems) -> None:
    """problems: up to four (X [K, M] bf16, dY [K, N] bf16, M, N, K, grad [M, N] f32) tuples -> grad += X^t dY for
    each, in ONE launch (no split-K atomics).  The four weight gradients of a GPT-2 block."""
    n = len(problems)
    arr = (GemmArgs * n)()
    for a, (X, dY, M, N, K, gout) in zip(arr, problems):
        a.A, a.B = X.data_ptr(), dY.data_ptr()
        a.M, a.N, a.K = M, N, K
        a.lda, a.ldb = M, N
        a.layout, a.epilogue, a.alpha = TN, EPI_NONE, 1.0
        a.out_f32, a.ld_out_f32 = gout.data_ptr(), N
        a.ld_out_bf16 = a.ld_res = a.ld_aux = N
        a.accumulate = 1
    _check(load().pgca_gemm_bf16_grouped(arr, n, _stream()), "pgca_gemm_bf16_grouped")


def gemm_skinny(x, W, M, N, K, scratch, *, lda=None, ldw=None, bias=None, act=EPI_NONE, residual=None, ld_res=None,
                out_f32=None, ld_out_f32=None, out_bf16=None, ld_out_bf16=None, ln=None, ln_out=None, ld_ln=None):
    """y = epilogue(x[M,K] . W[K,N]) for a handful of rows (incremental decoding); ``ln`` = (gamma, beta, eps) runs the
    next LayerNorm on the finished row into ``ln_out``.  ``scratch``: ``gemm_skinny_workspace(M, N, K)`` bytes."""
    a = SkinnyArgs()
    a.x, a.W = x.data_ptr(), W.data_ptr()
    a.M, a.N, a.K = M, N, K
    a.lda, a.ldw = (K if lda is None else lda), (N if ldw is None else ldw)
    a.scratch, a.bias, a.act = scratch.data_ptr(), _p(bias), act
    a.residual, a.ld_res = _p(residual), (N if ld_res is None else ld_res)
    a.out_f32, a.ld_out_f32 = _p(out_f32), (N if ld_out_f32 is None else ld_out_f32)
    a.out_bf16, a.ld_out_bf16 = _p(out_bf16), (N if ld_out_bf16 is None else ld_out_bf16)
    if ln is not None:
        a.ln_gamma, a.ln_beta, a.ln_eps = ln[0].data_ptr(), ln[1].data_ptr(), float(ln[2])
        a.ln_out_bf16, a.ld_ln = ln_out.data_ptr(), (N if ld_ln is None else ld_ln)
    _check(load().pgca_gemm_skinny(C.byref(a), _stream()), "pgca_gemm_skinny")


def gemm_skinny_workspace(M: int, N: int, K: int) -> int:
    return int(load().pgca_gemm_skinny_workspace(M, N, K))


def drop_args(seed: int, p: float):
    """(seed, threshold, scale) triple understood by every kernel with fused dropout; None when p == 0."""
    if p <= 0.0:
        return None
    return (seed & 0xFFFFFFFF, min(0xFFFFFFFF, int(p * 4294967296.0)), 1.0 / (1.0 - p))


def set_option(name: str, value: int) -> None:
    """Process-wide dispatch knob of the library (include/pgca_hip.h: pgca_set_option)."""
    _check(load().pgca_set_option(name.encode(), int(value)), "pgca_set_option")


gemm_probe = None  # optional object with want(layout, epilogue, M, N, K) / add(ev0, ev1, flops)


def rowstats_combine(stat_max, stat_sum, stat_ld, nparts, target_val, M, lse=None, out_logprob=None):
    _check(load().pgca_rowstats_combine(_p(stat_max), _p(stat_sum), stat_ld, nparts, _p(target_val), M, _p(lse),
                                        _p(out_logprob), _stream()), "pgca_rowstats_combine")


# --------------------------------------------------------------------------- LayerNorm / column sums
def layernorm_fwd(x, M, H, gamma, beta, eps=1e-5, row_map=None, y_bf16=None, y_f32=None, mean=None, rstd=None):
    _check(load().pgca_layernorm_fwd(_p(x), _p(row_map), M, H, _p(gamma), _p(beta), eps, _p(y_bf16), _p(y_f32),
                                     _p(mean), _p(rstd), _stream()), "pgca_layernorm_fwd")


def layernorm_bwd_blocks(M: int) -> int:
    return load().pgca_layernorm_bwd_blocks(M)


def layernorm_bwd(x, M, H, gamma, mean, rstd, dx_out, *, dy_bf16=None, dy_f32=None, row_map=None, add_to=None,
                  dx_bf16=None, part=None, part_extra=None, drop_add=None, drop_dx=None, drop_rows=None):
    _check(load().pgca_layernorm_bwd(_p(dy_bf16), _p(dy_f32), _p(x), _p(row_map), M, H, _p(gamma), _p(mean),
                                     _p(rstd), _p(add_to), _p(dx_out), _p(dx_bf16), _p(part), _p(part_extra),
                                     _drop_words(drop_add), _drop_words(drop_dx), _p(drop_rows), _stream()),
           "pgca_layernorm_bwd")


def colsum_finish(part, nparts, H, out, accumulate=False):
    _check(load().pgca_colsum_finish(_p(part), nparts, H, _p(out), 1 if accumulate else 0, _stream()),
           "pgca_colsum_finish")


def colsum_finish4(part, nplanes, nparts, H, outs, accumulate=False):
    o = list(outs) + [None] * (4 - len(outs))
    _check(load().pgca_colsum_finish4(_p(part), nplanes, nparts, H, _p(o[0]), _p(o[1]), _p(o[2]), _p(o[3]),
                                      1 if accumulate else 0, _stream()), "pgca_colsum_finish4")


def colsum_blocks(M: int) -> int:
    return load().pgca_colsum_blocks(M)


def colsum(M, N, ld, part, x_bf16=None, x_f32=None):
    _check(load().pgca_colsum(_p(x_bf16), _p(x_f32), M, N, ld, _p(part), _stream()), "pgca_colsum")


# --------------------------------------------------------------------------- attention
_NODROP = (0, 0, 1.0)


def _drop_words(d):
    """(seed, threshold, scale) -> host uint32[3] with the scale's float bits (NULL when dropout is off)."""
    if d is None:
        return None
    import struct
    arr = (C.c_uint32 * 3)(d[0], d[1], struct.unpack("<I", struct.pack("<f", d[2]))[0])
    return C.cast(arr, C.c_void_p)


def attention_fwd(qkv, key_mask, B, S, heads, causal, out, lse=None, drop=None, cu=None):
    """``cu`` (int32 [B+1]): packed rows - sequence b is rows cu[b]..cu[b+1]-1; S stays the padded length."""
    d = drop or _NODROP
    _check(load().pgca_attention_fwd(_p(qkv), _p(key_mask), B, S, heads, 1 if causal else 0, _p(out), _p(lse),
                                     d[0], d[1], d[2], _p(cu), _stream()), "pgca_attention_fwd")


def attention_bwd(qkv, out, dout, lse, key_mask, B, S, heads, causal, dqkv, drop=None, cu=None):
    d = drop or _NODROP
    _check(load().pgca_attention_bwd(_p(qkv), _p(out), _p(dout), _p(lse), _p(key_mask), B, S, heads,
                                     1 if causal else 0, _p(dqkv), d[0], d[1], d[2], _p(cu), _stream()),
           "pgca_attention_bwd")


# --------------------------------------------------------------------------- embeddings / ViT input
def embed_fwd(ids, B, S, H, wte, wpe, h0, attended=None, gamma=None, beta=None, eps=1e-5, mean=None, rstd=None,
              att_stride=None, U=None, xheads=0, drop_x=None, drop_e=None, row_ids=None, n_rows=0):
    """``row_ids`` (int32 [n_rows]): packed rows - output row r is padded position row_ids[r] (< 0: zero filler)."""
    _check(load().pgca_embed_fwd(_p(ids), B, S, H, _p(wte), _p(wpe), _p(attended), _p(gamma), _p(beta), eps, _p(h0),
                                 _p(mean), _p(rstd), H if att_stride is None else att_stride, _p(U), xheads,
                                 _drop_words(drop_x), _drop_words(drop_e), _p(row_ids), n_rows, _stream()),
           "pgca_embed_fwd")


def embed_bwd_blocks(B: int, S: int) -> int:
    return load().pgca_embed_bwd_blocks(B, S)


def embed_bwd(g, ids, row_mask, B, S, H, dwte, dwpe, wte=None, attended=None, gamma=None, mean=None, rstd=None,
              dattended=None, part=None, att_stride=None, U=None, dU=None, xheads=0, drop_x=None, drop_e=None,
              cu=None):
    _check(load().pgca_embed_bwd(_p(g), _p(ids), _p(row_mask), B, S, H, _p(wte), _p(attended), _p(gamma), _p(mean),
                                 _p(rstd), _p(dwte), _p(dwpe), _p(dattended), _p(part),
                                 H if att_stride is None else att_stride, _p(U), _p(dU), xheads, _drop_words(drop_x),
                                 _drop_words(drop_e), _p(cu), _stream()), "pgca_embed_bwd")


def patchify(pixels, B, image, patch, out_bf16, ld_out=None):
    ld = 3 * patch * patch if ld_out is None else ld_out
    _check(load().pgca_patchify(_p(pixels), B, image, patch, ld, _p(out_bf16), _stream()), "pgca_patchify")


def image_preprocess(images_u8, B, H, W, S, xbounds, xcoef, ybounds, ycoef, mean, std, tmp, out, resized_u8=None):
    _check(load().pgca_image_preprocess(_p(images_u8), B, H, W, S, _p(xbounds), _p(xcoef), xcoef.shape[1], _p(ybounds),
                                        _p(ycoef), ycoef.shape[1], mean[0], mean[1], mean[2], std[0], std[1], std[2],
                                        _p(tmp), _p(resized_u8), _p(out), _stream()), "pgca_image_preprocess")


def image_train_transform(images_u8, B, H, W, S, params, factors, xbounds, xcoef, ybounds, ycoef, mean, std, tmp, resized_u8,
                          out, aug_u8=None):
    _check(load().pgca_image_train_transform(_p(images_u8), B, H, W, S, _p(params), _p(factors), _p(xbounds), _p(xcoef),
                                             xcoef.shape[-1], _p(ybounds), _p(ycoef), ycoef.shape[-1], mean[0], mean[1],
                                             mean[2], std[0], std[1], std[2], _p(tmp), _p(resized_u8), _p(aug_u8), _p(out),
                                             _stream()), "pgca_image_train_transform")


def vit_assemble(patch_embeds, cls, pos, B, T, H, x):
    _check(load().pgca_vit_assemble(_p(patch_embeds), _p(cls), _p(pos), B, T, H, _p(x), _stream()),
           "pgca_vit_assemble")


def vit_assemble_bwd(dx, B, T, H, dpatch_bf16, dcls, dpos):
    _check(load().pgca_vit_assemble_bwd(_p(dx), B, T, H, _p(dpatch_bf16), _p(dcls), _p(dpos), _stream()),
           "pgca_vit_assemble_bwd")


# --------------------------------------------------------------------------- sequence reduce / losses
def seq_reduce(tok_lp, seq_of_row, nrows, nseq, seq_count, mode, seq_lp):
    _check(load().pgca_seq_reduce(_p(tok_lp), _p(seq_of_row), nrows, nseq, _p(seq_count), mode, _p(seq_lp),
                                  _stream()), "pgca_seq_reduce")


def logits_logprob_bwd(logits, ld, V, row_map, targets, g, R, dlogits):
    _check(load().pgca_logits_logprob_bwd(_p(logits), ld, V, _p(row_map), _p(targets), _p(g), R, _p(dlogits), _stream()),
           "pgca_logits_logprob_bwd")


def logits_logprob(logits, ld, V, row_map, targets, R, out):
    _check(load().pgca_logits_logprob(_p(logits), ld, V, _p(row_map), _p(targets), R, _p(out), _stream()),
           "pgca_logits_logprob")


def dpo_loss(pol_w, pol_l, ref_w, ref_l, B, beta, label_smoothing, loss, dpol_w=None, dpol_l=None, metrics=None):
    _check(load().pgca_dpo_loss(_p(pol_w), _p(pol_l), _p(ref_w), _p(ref_l), B, beta, label_smoothing, _p(loss),
                                _p(dpol_w), _p(dpol_l), _p(metrics), _stream()), "pgca_dpo_loss")


def seq_batch_prepare(ids, mask, Bq, S, counts, mask32, row_map, targets, seq_of_row, n_rows, stream=None):
    _check(load().pgca_seq_batch_prepare(_p(ids), _p(mask), Bq, S, _p(counts), _p(mask32), _p(row_map), _p(targets),
                                         _p(seq_of_row), _p(n_rows), _stream() if stream is None else stream),
           "pgca_seq_batch_prepare")


def seq_pack_prepare(mask32, Bq, S, pad_to, lens, cu, row_ids, n_packed, counts=None, row_map=None):
    _check(load().pgca_seq_pack_prepare(_p(mask32), Bq, S, pad_to, _p(lens), _p(cu), _p(row_ids), _p(n_packed),
                                        _p(counts), _p(row_map), _stream()), "pgca_seq_pack_prepare")


def row_scale(dseq, seq_of_row, seq_count, nrows, mode, out):
    _check(load().pgca_row_scale(_p(dseq), _p(seq_of_row), _p(seq_count), nrows, mode, _p(out), _stream()),
           "pgca_row_scale")


def masked_mean_fwd(feats, mask, B, S, H, pooled, cu=None):
    _check(load().pgca_masked_mean_fwd(_p(feats), _p(mask), B, S, H, _p(pooled), _p(cu), _stream()),
           "pgca_masked_mean_fwd")


def masked_mean_bwd(dpooled, mask, B, S, H, dfeats, cu=None):
    _check(load().pgca_masked_mean_bwd(_p(dpooled), _p(mask), B, S, H, _p(dfeats), _p(cu), _stream()),
           "pgca_masked_mean_bwd")


def l2norm_fwd(x, B, P, y, norm=None):
    _check(load().pgca_l2norm_fwd(_p(x), B, P, _p(y), _p(norm), _stream()), "pgca_l2norm_fwd")


def l2norm_bwd(dy, y, norm, B, P, dx):
    _check(load().pgca_l2norm_bwd(_p(dy), _p(y), _p(norm), B, P, _p(dx), _stream()), "pgca_l2norm_bwd")


def ntxent_loss(lse_r, lse_c, diag, n_local, n_total, loss):
    _check(load().pgca_ntxent_loss(_p(lse_r), _p(lse_c), _p(diag), n_local, n_total, _p(loss), _stream()),
           "pgca_ntxent_loss")


# --------------------------------------------------------------------------- optimiser
def sqnorm_blocks(n: int) -> int:
    return load().pgca_sqnorm_blocks(n)


def sqnorm(g, n, part):
    _check(load().pgca_sqnorm(_p(g), n, _p(part), _stream()), "pgca_sqnorm")


def step_control(part, nparts, max_norm, base_lr, warmup, total_steps, sched_stride, beta1, beta2, grad_scale, ctrl,
                 gate=None):
    _check(load().pgca_step_control(_p(part), nparts, max_norm, base_lr, warmup, total_steps, sched_stride, beta1,
                                    beta2, grad_scale, _p(gate), _p(ctrl), _stream()), "pgca_step_control")


def clip_coef(part, nparts, max_norm, coef):
    _check(load().pgca_clip_coef(_p(part), nparts, max_norm, _p(coef), _stream()), "pgca_clip_coef")


def scale_dev(x, n, coef):
    _check(load().pgca_scale_dev(_p(x), n, _p(coef), _stream()), "pgca_scale_dev")


def adamw(p, g, m, v, p_bf16, n, ctrl, weight_decay, beta1, beta2, eps, grad_scale=1.0):
    _check(load().pgca_adamw(_p(p), _p(g), _p(m), _p(v), _p(p_bf16), n, _p(ctrl), weight_decay, beta1, beta2, eps,
                             grad_scale, _stream()), "pgca_adamw")


def cast_bf16(x, y, n):
    _check(load().pgca_cast_bf16(_p(x), _p(y), n, _stream()), "pgca_cast_bf16")


def cast_f32(x_bf16, y, n):
    _check(load().pgca_cast_f32(_p(x_bf16), _p(y), n, _stream()), "pgca_cast_f32")


def split_bf16(x, R, P, rows_out, pattern, y):
    _check(load().pgca_split_bf16(_p(x), R, P, rows_out, pattern, _p(y), _stream()), "pgca_split_bf16")


def axpy(x, alpha, y, n, accumulate=False):
    _check(load().pgca_axpy(_p(x), alpha, _p(y), n, 1 if accumulate else 0, _stream()), "pgca_axpy")


def gather_rows_bf16(src, row_map, M, H, dst):
    _check(load().pgca_gather_rows_bf16(_p(src), _p(row_map), M, H, _p(dst), _stream()), "pgca_gather_rows_bf16")
