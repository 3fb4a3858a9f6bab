/*
 * pgca_hip.h - C ABI of libpgca_hip.so: the MI355X (gfx950) kernels behind the
 * Stage-1 NT-Xent and Stage-2 DPO steps of preference-guided captioning alignment.
 *
 * The reference (A-SHOJAEI/preference-guided-image-captioning-alignment) is 100 %
 * Python: it has no FFI of its own.  Its seam for this path is the Python surface
 * (models/model.py, models/components.py, training/trainer.py); each entry point
 * below names the reference call site whose ATen/cuBLAS/SDPA launches it replaces.
 * The binding a maintainer adds on the reference side is a ctypes stub - see
 * INTEGRATION.md.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless
 *     stated otherwise; buffers are caller-owned (the Python host allocates them
 *     with torch's caching allocator and passes tensor.data_ptr()).
 *   - `stream` is a hipStream_t passed as void*; kernels are asynchronous on it,
 *     never allocate, never synchronise, keep no global mutable state except the
 *     thread-local last-error string.
 *   - bf16 = IEEE bfloat16 stored as uint16_t; f32 accumulation everywhere.
 *   - token ids / targets are int64 exactly as the reference supplies them.
 *   - return value: 0 on success, negative pgca_status otherwise.
 */
#ifndef PGCA_HIP_H
#define PGCA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum pgca_status {
  PGCA_OK = 0,
  PGCA_ERR_INVALID = -1, /* bad shape / alignment / null pointer */
  PGCA_ERR_LAUNCH = -2   /* hipLaunch failed; see pgca_last_error() */
} pgca_status;

#define PGCA_ABI_VERSION 304 /* bumped whenever a signature or struct layout below changes */
int pgca_version(void);        /* == PGCA_ABI_VERSION of the header the library was built from */
int pgca_sizeof_gemm_args(void); /* sizeof(pgca_gemm_args) as compiled: bindings compare it with their own layout */
const char* pgca_last_error(void);

/* ------------------------------------------------------------------ GEMM (MFMA bf16) */
/* Operand layouts.  "K" is the contraction dimension.
 *   PGCA_NT: A[M,K] row-major, B[N,K] row-major   (nn.Linear forward; dgrad through Conv1D)
 *   PGCA_NN: A[M,K] row-major, B[K,N] row-major   (GPT-2 Conv1D forward; dgrad through nn.Linear)
 *   PGCA_TN: A[K,M] row-major, B[K,N] row-major   (every weight gradient: X^t dY)
 * Constraints: lda/ldb multiples of 8 elements, base pointers 16-B aligned; for
 * operands whose contiguous dimension is K (A in NT/NN, B in NT) K % 8 == 0. */
enum { PGCA_NT = 0, PGCA_NN = 1, PGCA_TN = 2 };

/* Epilogues, applied to v = alpha*acc (+ bias[n]). */
enum {
  PGCA_EPI_NONE = 0,
  PGCA_EPI_GELU_NEW = 1,   /* GPT-2 MLP (modeling_gpt2.py:229-243); aux_out (opt.) <- pre-activation */
  PGCA_EPI_QUICK_GELU = 2, /* CLIP MLP (activations.py:117-123) */
  PGCA_EPI_RELU = 3,       /* projection heads (model.py:138,340) */
  PGCA_EPI_TANH = 4,       /* vision_projection (model.py:523) */
  PGCA_EPI_DGELU_NEW = 5,  /* v *= gelu_new'(aux_in[m,n]) (aux_in = saved pre-activation) */
  PGCA_EPI_DRELU = 6,      /* v *= aux_in[m,n] > 0       (aux_in = saved activation) */
  PGCA_EPI_DTANH = 7,      /* v *= 1 - aux_in[m,n]^2     (aux_in = saved activation) */
  PGCA_EPI_ROWSTATS = 8,   /* no C written: per-row partial (max, sum exp) over this block's columns
                              + target-logit pick; fused LM head / NT-Xent (model.py:1069-1079,988-998) */
  PGCA_EPI_DLOGITS = 9,    /* v = row_scale[m] * (exp(v - row_lse[m]) - (n == target[m])); 0 for n >= N;
                              aux_out (opt., bf16 [M, ld_aux]) <- bf16(v - bf16(v)): low half of a hi/lo split */
  PGCA_EPI_DQUICK_GELU = 10, /* v *= quick_gelu'(aux_in[m,n]) (aux_in = saved pre-activation; trainable CLIP tower) */
  /* The GPT-2 MLP pair that evaluates the sigmoid ONCE: the forward epilogue has it in registers for gelu_new and writes
   * gelu_new'(pre) beside the activation; the backward epilogue is then a plain multiply instead of exp + rcp + 8 FMAs
   * per element on the K = 1024 data-gradient GEMM whose epilogue was 40 % of its tile life. */
  PGCA_EPI_GELU_NEW_D = 11,  /* like GELU_NEW, but aux_out <- gelu_new'(pre-activation) (bf16) */
  PGCA_EPI_MUL_AUX = 12      /* v *= aux_in[m,n] (aux_in = the derivative saved by GELU_NEW_D); colsum_part allowed */
};

typedef struct pgca_gemm_args {
  const void* A;  /* bf16 */
  const void* B;  /* bf16 */
  int32_t M, N, K;
  int32_t lda, ldb;
  int32_t layout;
  int32_t epilogue;
  float alpha;            /* scale on the accumulator (1/tau for NT-Xent) */
  const float* bias;      /* [N] f32 or NULL */
  void* out_bf16;         /* [M, ld_out_bf16] or NULL */
  int32_t ld_out_bf16;
  float* out_f32;         /* [M, ld_out_f32] or NULL */
  int32_t ld_out_f32;
  int32_t accumulate;     /* out_f32 += v instead of = v (gradient accumulation) */
  const float* residual;  /* [M, ld_res] f32 added to v before the store, or NULL */
  int32_t ld_res;
  void* aux_out;          /* bf16 [M, ld_aux]: pre-activation written by *_GELU epilogues, or NULL */
  const void* aux_in;     /* bf16 [M, ld_aux]: operand of the D* epilogues */
  int32_t ld_aux;
  /* ROWSTATS / DLOGITS */
  const int64_t* targets; /* [M] column index per row, or NULL (NT-Xent: diagonal given explicitly) */
  float* stat_max;        /* [M, stat_ld]  partial row max   (ROWSTATS) */
  float* stat_sum;        /* [M, stat_ld]  partial sum exp(v - max) */
  int32_t stat_ld;        /* >= 2 * ceil(N / 128) */
  float* target_val;      /* [M] v at the target column (ROWSTATS) */
  const float* row_lse;   /* [M] (DLOGITS) */
  const float* row_scale; /* [M] (DLOGITS) */
  int32_t out_cols;       /* DLOGITS: columns written (>= N, padding columns get 0) */
  /* fused dropout on the epilogue value (after the activation / derivative, before the residual add):
   * element (m, n) is multiplied by 0 or drop_scale according to pgca's counter hash of (drop_seed, m*N + n);
   * drop_threshold = p * 2^32, 0 disables.  Replaces nn.Dropout at model.py:139,341,524 and GPT-2 resid_dropout
   * (modeling_gpt2.py:224,242); the backward passes the same seed to replay the mask. */
  uint32_t drop_seed;
  uint32_t drop_threshold;
  float drop_scale;
  /* Optional, PGCA_EPI_DGELU_NEW / PGCA_EPI_MUL_AUX only: f32 [ceil(M/64), ld_colsum] - row b receives the column sums of the rows
   * 64b .. 64b+63 of the result (before its bf16 rounding); summed over b (pgca_colsum_finish) they are the bias gradient
   * of the layer whose pre-activation gradient this GEMM produces, without a second pass over the M x N result. */
  float* colsum_part;
  int32_t ld_colsum;
  /* Optional, packed (variable-length) token rows: drop_rows[m] is the row index the dropout hash is keyed on instead of m
   * (the token's position b*S + t in the PADDED [B, S] layout of the reference batch), so a packed launch draws exactly the
   * masks of the padded one.  int32 [M]; entries of padding rows may hold anything (their values are never used). */
  const int32_t* drop_rows;
} pgca_gemm_args;

int pgca_gemm_bf16(const pgca_gemm_args* args, void* stream);
/* Which kernel pgca_gemm_bf16 would launch for these arguments: schedule*1000000 + tile*100 + K-splits
 * (12801 = general 128^2 register-staged kernel; 256xx = 256^2 LDS-DMA tile with xx K-splits; schedule
 * 0 = 2-stage BK=64 loop (gemm256_kernel, default for the K-strided TN layout), 6 = phase-staggered 4-stage BK=32
 * loop (gemm256s_kernel, default for NT / NN)). */
int pgca_gemm_plan(const pgca_gemm_args* args);
/* Process-wide dispatch knobs (tests, micro-benchmarks, tuning).  Defaults come from the environment, read ONCE at
 * first use - no launch calls getenv:
 *   "gemm_tile"      0 automatic | 128 | 256                          (PGCA_GEMM_TILE)
 *   "gemm_schedule"  -1 automatic | 0 | 6 (see pgca_gemm_plan)         (PGCA_GEMM_RING)
 *   "gemm_group"     1 grouped weight-gradient launch | 0 one by one  (PGCA_GEMM_NO_GROUP=1 -> 0)
 *   "gemm_stagger"   0..64: start-delay step (x 1024 clocks) of the first wave of workgroups of a many-round
 *                    gemm256s launch, which de-phases the CUs' epilogue store bursts   (PGCA_GEMM_STAGGER)
 * Returns PGCA_ERR_INVALID for an unknown name or value. */
int pgca_set_option(const char* name, int32_t value);
/* `count` GEMMs in one launch.  Up to four TN problems with epilogue NONE and an f32 (accumulating) output - the four
 * weight gradients of one GPT-2 block (autograd of Conv1D c_attn / c_proj / c_fc / mlp.c_proj, reference
 * modeling_gpt2.py:203,222-224,229-243 under trainer.py:494,606 loss.backward()) - run as ONE grid with the whole K per
 * tile: no split-K, no atomics.  Anything else falls back to `count` ordinary pgca_gemm_bf16 launches. */
int pgca_gemm_bf16_grouped(const pgca_gemm_args* args, int32_t count, void* stream);

/* Skinny product for incremental decoding (generation with a K/V cache: reference models/model.py:657-675 -> HF generate,
 * modeling_gpt2.py:144-226 with layer_past): y[M, N] = epilogue(x[M, K] . W[K, N]) for 1 <= M <= PGCA_SKINNY_MAX_M rows,
 * W row-major [K, N] (GPT-2 Conv1D), one pass over W spread over every CU (column chunks x K splits, each storing its partial
 * [M, N] slab in `scratch`), then a per-row finish that sums the slabs in a fixed order (bitwise reproducible):
 * v = acc + bias[n]; act (PGCA_EPI_NONE | PGCA_EPI_GELU_NEW); + residual[m, n];
 * out_f32 / out_bf16 (either or both, arbitrary row strides - e.g. a K/V-cache row); optionally the LayerNorm of the finished
 * f32 row (ln_gamma/ln_beta/ln_eps -> ln_out_bf16 [M, ld_ln]: the operand of the next product, no separate LN launch).
 * scratch: pgca_gemm_skinny_workspace(M, N, K) bytes (16-B aligned), contents irrelevant on entry.
 * Constraints: N % 8 == 0, N <= 8192, ldw % 8 == 0, pointers 16-B aligned, row strides multiples of 4 elements. */
#define PGCA_SKINNY_MAX_M 64
typedef struct pgca_skinny_args {
  const void* x;   /* bf16 [M, lda] */
  const void* W;   /* bf16 [K, ldw] */
  int32_t M, N, K, lda, ldw;
  float* scratch;
  const float* bias;      /* [N] or NULL */
  int32_t act;
  const float* residual;  /* f32 [M, ld_res] or NULL */
  int32_t ld_res;
  float* out_f32;
  int32_t ld_out_f32;
  void* out_bf16;
  int32_t ld_out_bf16;
  const float* ln_gamma;
  const float* ln_beta;
  float ln_eps;
  void* ln_out_bf16;      /* NULL: no fused LayerNorm */
  int32_t ld_ln;
} pgca_skinny_args;
int pgca_gemm_skinny(const pgca_skinny_args* args, void* stream);
int64_t pgca_gemm_skinny_workspace(int32_t M, int32_t N, int32_t K);
int pgca_sizeof_skinny_args(void);

/* Per-row combine of ROWSTATS partials: lse[m] = log sum exp over all columns;
 * out_logprob[m] = target_val[m] - lse[m] (token log-prob, reference model.py:1074-1079). */
int pgca_rowstats_combine(const float* stat_max, const float* stat_sum, int32_t stat_ld, int32_t nparts,
                          const float* target_val, int32_t M, float* lse, float* out_logprob, void* stream);

/* ------------------------------------------------------------------ LayerNorm */
/* y = LN(x) * gamma + beta, eps as given (1e-5 everywhere in the reference).
 * x f32 [rows_in, H]; optional row_map[M] gathers rows (y row m <- x row row_map[m]);
 * y_bf16 / y_f32 [M, H] (either may be NULL); mean/rstd [M] saved for backward (may be NULL).
 * Replaces F.layer_norm at model.py:141,343,535,601 and HF ln_1/ln_2/ln_f, CLIP layer_norm*. */
int pgca_layernorm_fwd(const float* x, const int32_t* row_map, int32_t M, int32_t H, const float* gamma,
                       const float* beta, float eps, void* y_bf16, float* y_f32, float* mean, float* rstd,
                       void* stream);

/* dx = LN backward; dy given as bf16 (dy_bf16) or f32 (dy_f32), exactly one non-NULL.
 * dx_out[row] = (add_to ? add_to[row] : 0) + dx   with row = row_map ? row_map[m] : m.
 * dx_bf16 (optional) receives the same value rounded to bf16 (operand of the next dgrad GEMM).
 * dgamma/dbeta partial sums go to part[2, nblk, H] (nblk returned by pgca_layernorm_bwd_blocks);
 * pgca_colsum_finish folds them into the gradient buffers.  part_extra[2, nblk, H] (optional) receives the column
 * sums of add_to and of dx_out - the bias gradients of the two GEMMs around this LayerNorm, for free.
 * drop_add / drop_dx (HOST pointers to {seed, threshold, scale-as-float-bits}, or NULL) replay the dropout masks of
 * those two GEMM outputs: the add_to column sum uses drop_add; dx_bf16 and the dx column sum use drop_dx (dx_out
 * itself, the f32 residual-stream gradient, is never masked).  drop_rows (optional, int32 [rows]): the row index the two
 * dropout hashes are keyed on (see pgca_gemm_args::drop_rows; packed token rows). */
int pgca_layernorm_bwd_blocks(int32_t M);
int pgca_layernorm_bwd(const void* dy_bf16, const float* dy_f32, const float* x, const int32_t* row_map,
                       int32_t M, int32_t H, const float* gamma, const float* mean, const float* rstd,
                       const float* add_to, float* dx_out, void* dx_bf16, float* part, float* part_extra,
                       const uint32_t* drop_add, const uint32_t* drop_dx, const int32_t* drop_rows, void* stream);
/* out[h] (+)= sum_b part[b, h]; nparts rows of length H. */
int pgca_colsum_finish(const float* part, int32_t nparts, int32_t H, float* out, int32_t accumulate, void* stream);

/* Same for up to four planes part[plane][nparts][H] -> out0..out3 in one launch. */
int pgca_colsum_finish4(const float* part, int32_t nplanes, int32_t nparts, int32_t H, float* out0, float* out1,
                        float* out2, float* out3, int32_t accumulate, void* stream);

/* Column sums of a bf16 / f32 matrix (bias gradients): out[n] (+)= sum_m x[m, n]. */
int pgca_colsum_blocks(int32_t M);
int pgca_colsum(const void* x_bf16, const float* x_f32, int32_t M, int32_t N, int32_t ld, float* part, void* stream);

/* ------------------------------------------------------------------ attention (head_dim 64) */
/* qkv bf16 [B*S, 3*H] (q | k | v, head h at columns h*64), out bf16 [B*S, H], lse f32 [B, heads, S].
 * softmax(q k^t / 8 + mask) v with mask = causal AND key_mask[b, key] != 0 (key_mask int32 [B,S] or NULL).
 * Any S in the forward (one on-chip tile up to 128, key-tiled online softmax beyond); the backward keeps the dQ
 * accumulators of every 128-query block in registers: S <= PGCA_ATTN_MAX_S.  B <= 65535.
 * Replaces SDPA at modeling_gpt2.py:54-72,203-215 and modeling_clip.py:259-277.
 * Packed (variable-length) rows: with cu_seqlens (int32 [B+1], cu[0] = 0) sequence b occupies rows cu[b] .. cu[b+1]-1 of
 * qkv / out / dout / dqkv and has min(cu[b+1]-cu[b], S) tokens (a KV-cache decode step passes the cache stride in cu and
 * the filled length as S, with no mask, lse or dropout); in training S stays the PADDED length: key_mask, lse and the dropout index
 * keep their [B, S] geometry, so the result of every real token equals the padded launch's.  Padding positions of the
 * reference batch (model.py:1069-1083 zeroes their loss terms, :449-456 their pooling weight) are then never computed. */
#define PGCA_ATTN_MAX_S 512
int pgca_attention_fwd(const void* qkv, const int32_t* key_mask, int32_t B, int32_t S, int32_t heads,
                       int32_t causal, void* out, float* lse, uint32_t drop_seed, uint32_t drop_threshold,
                       float drop_scale, const int32_t* cu_seqlens, void* stream);
/* dqkv bf16 [B*S, 3*H] from dout bf16 [B*S, H], the saved qkv / out / lse.  drop_*: attention-probability
 * dropout (element index ((b*heads + h)*S + q)*S + key; threshold 0 disables), replayed in the backward. */
int pgca_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse,
                       const int32_t* key_mask, int32_t B, int32_t S, int32_t heads, int32_t causal,
                       void* dqkv, uint32_t drop_seed, uint32_t drop_threshold, float drop_scale,
                       const int32_t* cu_seqlens, void* stream);

/* ------------------------------------------------------------------ embeddings */
/* Caption-decoder input (reference model.py:591-601 + modeling_gpt2.py:571-577):
 *   e = wte[ids[b,s]] + attended[b]; x = LN(e; attention_norm); h0[b,s] = x + wpe[s]
 * attended[b] = W_o(W_v pv_b + b_v) + b_o is the collapsed 1-key cross-attention.
 * Text tower (modeling_gpt2.py:568-577): attended == NULL and gamma == NULL -> h0 = wte[id] + wpe[s].
 * mean/rstd [B*S] saved when LN is applied. ids int64. */
/* Train-mode extras (all optional): attended is read at row b*att_stride (0 = one shared vector, i.e. b_o);
 * U [B, xheads, H] adds sum_h w(b,h,s) U[b,h,:] with w the replayable dropout multiplier of the 1-key
 * attention weight (drop_x, index (b*xheads + h)*S + s); drop_e is GPT-2's embedding dropout on h0
 * (index m*H + c).  drop_* are HOST pointers to {seed, threshold, scale bits} or NULL.
 * Packed rows: with row_ids (int32 [n_rows]) output row r (and mean/rstd[r]) is position m = row_ids[r] = b*S + t of the
 * padded batch (ids, positions and every dropout index stay keyed on m); row_ids[r] < 0 marks a filler row, written as
 * zeros.  Without row_ids n_rows is ignored and all B*S rows are produced. */
int pgca_embed_fwd(const int64_t* ids, int32_t B, int32_t S, int32_t H, const float* wte, const float* wpe,
                   const float* attended, const float* gamma, const float* beta, float eps, float* h0,
                   float* mean, float* rstd, int32_t att_stride, const float* U, int32_t xheads,
                   const uint32_t* drop_x, const uint32_t* drop_e, const int32_t* row_ids, int32_t n_rows,
                   void* stream);
/* Backward of the above given g = dL/dh0 [B*S, H] (f32):
 *   dwpe[s] += sum_b g; through LN (if gamma) -> de; dwte[ids] += de (rows with row_mask==0 skipped:
 *   their gradient is exactly zero); dattended[b] = sum_s de; dgamma/dbeta partials in part[2,nblk,H].
 * Packed rows: with cu_seqlens (int32 [B+1]) g, mean and rstd hold position (b, s) at row cu[b] + s, s < cu[b+1]-cu[b];
 * positions beyond a sequence's packed length contribute nothing (they are padding: row_mask is 0 there). */
int pgca_embed_bwd(const float* g, const int64_t* ids, const int32_t* row_mask, int32_t B, int32_t S, int32_t H,
                   const float* wte, const float* attended, const float* gamma, const float* mean,
                   const float* rstd, float* dwte, float* dwpe, float* dattended, float* part, int32_t att_stride,
                   const float* U, float* dU, int32_t xheads, const uint32_t* drop_x, const uint32_t* drop_e,
                   const int32_t* cu_seqlens, void* stream);
int pgca_embed_bwd_blocks(int32_t B, int32_t S);

/* ViT patch gather (modeling_clip.py:200-218): pixels f32 [B,3,I,I] -> bf16 [B*G*G, ld_out] in the
 * (c, ky, kx) order of the conv weight, so the bias-free Conv2d is one NT GEMM.  ld_out >= 3*P*P; columns
 * 3*P*P..ld_out-1 are written as zeros (ViT-L/14: 588 -> 640, K of the GEMM padded to its 64-deep tile).  P even. */
int pgca_patchify(const float* pixels, int32_t B, int32_t image, int32_t patch, int32_t ld_out, void* out_bf16,
                  void* stream);
/* x[b, 0] = cls + pos[0]; x[b, 1+p] = patches[b, p] + pos[1+p]  (f32 [B, T, H]). */
int pgca_vit_assemble(const float* patch_embeds, const float* cls, const float* pos, int32_t B, int32_t T,
                      int32_t H, float* x, void* stream);
/* Backward of pgca_vit_assemble for a trainable tower (reference model.py:150-164 leaves the CLIP tower trainable unless
 * freeze_vision_backbone): dx f32 [B, T, H] -> dpatch bf16 [B, T-1, H] (operand of the patch-embedding weight gradient),
 * dpos[t] += sum_b dx[b, t], dcls += sum_b dx[b, 0]  (both ACCUMULATE, like every parameter gradient). */
int pgca_vit_assemble_bwd(const float* dx, int32_t B, int32_t T, int32_t H, void* dpatch_bf16, float* dcls,
                          float* dpos, void* stream);

/* Image input transform of the reference's loader on the device (data/preprocessing.py:44-48,78: Resize((S, S)) ->
 * ToTensor -> Normalize on a PIL RGB image), BIT-EXACT with the host path: images u8 [B, H, W, 3]
 * (decoded RGB, HWC) -> Pillow's two-pass antialiased bilinear resample on 8-bit channels (22-bit fixed-point taps,
 * round half up and clip after EACH pass; Resample.c) -> float32(v) / 255 -> (t - mean) / std -> out f32 [B, 3, S, S].
 * x/ybounds int32 [S, 2] = (first tap, tap count), x/ycoef int32 [S, xk / yk]: Pillow's precompute_coeffs +
 * normalize_coeffs_8bpc for W -> S and H -> S (host side: pgca_amd.input.resample_tables).  tmp u8 [B, H, S, 3] is the
 * intermediate of the horizontal pass; resized_u8 (optional) u8 [B, S, S, 3] receives the resized image itself. */
int pgca_image_preprocess(const uint8_t* images, int32_t B, int32_t H, int32_t W, int32_t S, const int32_t* xbounds,
                          const int32_t* xcoef, int32_t xk, const int32_t* ybounds, const int32_t* ycoef, int32_t yk,
                          float mean0, float mean1, float mean2, float std0, float std1, float std2, uint8_t* tmp,
                          uint8_t* resized_u8, float* out, void* stream);

/* TRAINING transform of the reference's loader on the device (data/preprocessing.py:52-70, augment=True:
 * RandomResizedCrop(S, scale (0.8, 1), ratio (0.75, 1.33)) -> RandomHorizontalFlip -> ColorJitter(0.2, 0.2, 0.2, 0.1) ->
 * RandomRotation(5) -> ToTensor -> Normalize), BIT-EXACT with torchvision's PIL backend GIVEN the random draws, which
 * stay on the host (pgca_amd.input.draw_train_params restates torchvision's get_params).  Per image b:
 *   params int32 [B, 20]: crop box i, j, h, w (top, left, height, width; inside the H x W image) | flip 0/1 |
 *     the four ColorJitter operations in their drawn order (0 brightness, 1 contrast, 2 saturation, 3 hue) |
 *     hue turn = uint8(hue_factor * 255) | rotate 0/1 (0: angle % 360 == 0, PIL copies) |
 *     a0 a1 a2 a3 a4 a5: Geometry.c affine_fixed's 16.16 integers of the output->input map | 3 unused;
 *   factors f32 [B, 3]: brightness, contrast, saturation (Blend.c alpha);
 *   x/ybounds int32 [B, S, 2], x/ycoef int32 [B, S, xk / yk]: resample taps for w_b -> S and h_b -> S (rows padded with
 *     zeros to the batch's widest table).
 * tmp u8 [B, H, S, 3] and resized_u8 u8 [B, S, S, 3] are scratch (the latter returns the cropped + resized image);
 * aug_u8 (optional) u8 [B, S, S, 3] receives the augmented image before ToTensor; out f32 [B, 3, S, S].
 * The jitter + rotation stage keeps one S x S x 3 image in LDS: S <= 230 (every shipped config uses 224). */
int pgca_image_train_transform(const uint8_t* images, int32_t B, int32_t H, int32_t W, int32_t S, const int32_t* params,
                               const float* factors, const int32_t* xbounds, const int32_t* xcoef, int32_t xk,
                               const int32_t* ybounds, const int32_t* ycoef, int32_t yk, float mean0, float mean1,
                               float mean2, float std0, float std1, float std2, uint8_t* tmp, uint8_t* resized_u8,
                               uint8_t* aug_u8, float* out, void* stream);

/* ------------------------------------------------------------------ sequence reduce + losses */
/* tok_lp f32 [nrows] are token log-probs of the COMPACT rows; row r belongs to sequence seq_of_row[r].
 * seq_lp[q] = sum (mode 0; components.py:357-362) or mean over the sequence's scored tokens
 * (mode 1; model.py:1082-1083; 0/0 = NaN when a caption has <= 1 real token, as the reference). */
int pgca_seq_reduce(const float* tok_lp, const int32_t* seq_of_row, int32_t nrows, int32_t nseq,
                    const int32_t* seq_count, int32_t mode, float* seq_lp, void* stream);
/* Token log-probs from MATERIALISED f32 logits (API-compatibility path of PreferenceLoss /
 * compute_sequence_logprobs, model.py:1069-1079, components.py:340-354):
 * out[r] = logits[row_map[r], targets[r]] - logsumexp(logits[row_map[r], 0:V]). */
int pgca_logits_logprob(const float* logits, int32_t ld, int32_t V, const int32_t* row_map, const int64_t* targets,
                        int32_t R, float* out, void* stream);
/* Backward of pgca_logits_logprob for callers that hold materialised logits WITH a gradient (the reference's loss objects
 * are differentiated by autograd: model.py:1069-1083, components.py:340-362): for each compact row r,
 * dlogits[row_map[r], 0:V] = g[r] * (onehot(targets[r]) - softmax(logits[row_map[r], 0:V])), g[r] = dLoss/d tok_lp[r];
 * rows that score no token are not written (the caller zero-fills dlogits). */
int pgca_logits_logprob_bwd(const float* logits, int32_t ld, int32_t V, const int32_t* row_map, const int64_t* targets,
                            const float* g, int32_t R, float* dlogits, void* stream);
/* DPO / preference loss over B pairs (components.py:210-231, model.py:1047-1048).
 * pol_w/pol_l/ref_w/ref_l f32 [B] (ref_* may be NULL = reference-free).
 * loss[0] = mean loss; dpol_w/dpol_l [B] = dLoss/dpol (f32), all zero when the loss is not finite (the reference skips
 * such a batch, trainer.py:606-613); metrics[4] = reward_margin, reward_accuracy, mean pol_w, mean pol_l
 * (components.py:234-247), all on device (no host sync). */
int pgca_dpo_loss(const float* pol_w, const float* pol_l, const float* ref_w, const float* ref_l, int32_t B,
                  float beta, float label_smoothing, float* loss, float* dpol_w, float* dpol_l, float* metrics,
                  void* stream);
/* The index work of the log-prob gather on the device (reference model.py:1069-1083, components.py:340-357: shift,
 * mask product, gather index), replacing the host pass of a collate function: for ids / mask int64 [Bq, S] keeps the
 * rows (b, t) with mask[b, t+1] != 0, sorted by sequence: row_map[r] = b*S + t, targets[r] = ids[b, t+1] (bit-exact),
 * seq_of_row[r] = b (each of capacity Bq*(S-1)); counts[b] = kept rows of sequence b; n_rows[0] = total;
 * mask32 (optional) = int32 0/1 copy of the mask (key mask of pgca_attention_*). */
int pgca_seq_batch_prepare(const int64_t* ids, const int64_t* mask, int32_t Bq, int32_t S, int32_t* counts,
                           int32_t* mask32, int32_t* row_map, int64_t* targets, int32_t* seq_of_row, int32_t* n_rows,
                           void* stream);
/* Packed (variable-length) row layout of a right-padded batch: only positions t < len[b] = 1 + (last t with mask[b,t] != 0)
 * are given a row, sequence after sequence (reference semantics: with a causal AND key-padding mask, a masked mean and a
 * masked loss - model.py:449-456,1069-1083, SURVEY 3.1 items 6-7 - nothing a padded position computes reaches a loss term or
 * a gradient).  With F = ceil((pad_to - 1) / S) filler pseudo-sequences (1 whenever S >= pad_to - 1):
 * mask32 int32 [Bq+F, S]: rows 0..Bq-1 as written by pgca_seq_batch_prepare; rows Bq.. are SET TO ONES here.
 *   lens[Bq]: the packed length of every sequence;
 *   cu[Bq+F+1]: cu[b] = first row of sequence b, cu[Bq] = n = sum len, cu[Bq+F] = n rounded up to a multiple of pad_to;
 *             rows cu[Bq]..cu[Bq+F]-1 are filler (extra unmasked "sequences" of zero embeddings, at most S rows each so
 *             that they obey the attention kernels' length bound) that keeps the row count a multiple of the GEMM K
 *             tile for the weight gradients;
 *   row_ids[cap]: row_ids[cu[b] + t] = b*S + t; filler rows get -1  (cap >= Bq*S rounded up to pad_to);
 *   n_packed[2] = {n, n rounded up};
 *   row_map (optional, in place, with counts from pgca_seq_batch_prepare): b*S + t  ->  cu[b] + t for the compact rows. */
int pgca_seq_pack_prepare(int32_t* mask32, int32_t Bq, int32_t S, int32_t pad_to, int32_t* lens, int32_t* cu,
                          int32_t* row_ids, int32_t* n_packed, const int32_t* counts, int32_t* row_map, void* stream);
/* row_scale[r] = +-dseq[seq_of_row[r]] * (mode & 1 ? 1/count : 1): dLoss/d tok_lp per compact row; mode & 2 negates
 * (the DLOGITS epilogue computes the cross-entropy form softmax - onehot, so it is fed -dLoss/dtok_lp). */
int pgca_row_scale(const float* dseq, const int32_t* seq_of_row, const int32_t* seq_count, int32_t nrows,
                   int32_t mode, float* row_scale, void* stream);

/* ------------------------------------------------------------------ pooling / normalise (Stage 1) */
/* pooled[b] = sum_s feats[b,s]*mask[b,s] / max(sum_s mask, 1)  (model.py:449-456); feats f32.
 * With cu_seqlens (int32 [B+1]) feats / dfeats are packed: position (b, s) is row cu[b] + s, s < cu[b+1]-cu[b]. */
int pgca_masked_mean_fwd(const float* feats, const int32_t* mask, int32_t B, int32_t S, int32_t H, float* pooled,
                         const int32_t* cu_seqlens, void* stream);
int pgca_masked_mean_bwd(const float* dpooled, const int32_t* mask, int32_t B, int32_t S, int32_t H,
                         float* dfeats, const int32_t* cu_seqlens, void* stream);
/* y = x / max(||x||, 1e-12) (F.normalize, model.py:828-829) and its backward. */
int pgca_l2norm_fwd(const float* x, int32_t B, int32_t P, float* y, float* norm, void* stream);
int pgca_l2norm_bwd(const float* dy, const float* y, const float* norm, int32_t B, int32_t P, float* dx,
                    void* stream);
/* NT-Xent pieces (model.py:988-998) on top of two ROWSTATS GEMMs (rows of S and rows of S^t):
 * loss[0] = (sum_i (lse_r[i] - diag[i]) + sum_i (lse_c[i] - diag[i])) / (2 * n_total) over n_local rows. */
int pgca_ntxent_loss(const float* lse_r, const float* lse_c, const float* diag, int32_t n_local, int32_t n_total,
                     float* loss, void* stream);

/* ------------------------------------------------------------------ optimiser (fused, flat buffers) */
/* Partial sums of squares of a flat f32 gradient buffer: part[blocks]; blocks = pgca_sqnorm_blocks(n). */
int pgca_sqnorm_blocks(int64_t n);
int pgca_sqnorm(const float* g, int64_t n, float* part, void* stream);
/* Device-side step control (no host sync): reads the partial sums of all segments, derives
 *   total_norm, finite flag, clip = min(1, max_norm/(norm+1e-6)) (clip_grad_norm_, trainer.py:511-515,619-623),
 * and - if finite - advances ctrl.step and evaluates the cosine warm-up lr (trainer.py:285-289).
 * ctrl layout (f32[8]): [0] total_norm [1] finite(1/0) [2] clip [3] lr [4] bias_corr1 [5] bias_corr2
 *                       [6] opt_step (as float) [7] sched_step (as float).
 * gate (device, optional): loss of the micro-batch that closes the accumulation group; non-finite -> the step is
 * skipped like a non-finite norm (the reference drops the group there, trainer.py:481-489 under accumulate()). */
int pgca_step_control(const float* part, int32_t nparts, float max_norm, float base_lr, int32_t warmup,
                      int32_t total_steps, int32_t sched_stride, float beta1, float beta2, float grad_scale,
                      const float* gate, float* ctrl, void* stream);
/* The reference's per-micro-batch clip_grad_norm_ (trainer.py:511-515,619-623) on a partially accumulated gradient,
 * without a host sync: coef[0] = min(1, max_norm / (norm + 1e-6)) and coef[1] = norm from the pgca_sqnorm partials of
 * all segments; pgca_scale_dev multiplies a flat buffer by coef[0]. */
int pgca_clip_coef(const float* part, int32_t nparts, float max_norm, float* coef, void* stream);
int pgca_scale_dev(float* x, int64_t n, const float* coef, void* stream);
/* AdamW (trainer.py:275-281: decoupled decay on every parameter) over a flat segment, gradient scaled by
 * ctrl.clip * grad_scale, skipped entirely when ctrl.finite == 0; refreshes the bf16 mirror. */
int pgca_adamw(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, const float* ctrl,
               float weight_decay, float beta1, float beta2, float eps, float grad_scale, void* stream);
/* f32 -> bf16 cast of a flat buffer (initial mirror / reference-policy snapshot). */
int pgca_cast_bf16(const float* x, void* y_bf16, int64_t n, void* stream);
/* bf16 -> f32 of a flat buffer (back from the bf16-compressed gradient all-reduce, dist.DataParallel). */
int pgca_cast_f32(const void* x_bf16, float* y, int64_t n, void* stream);
/* hi/lo bf16 split of an f32 matrix along K (NT-Xent similarity at f32-grade accuracy on the bf16 MFMA GEMM,
 * reference model.py:988-990 computes the similarity in fp32): x f32 [R, P] -> y bf16 [rows_out, 3P],
 * row r = [hi | hi | lo] (pattern 0, the A operand) or [hi | lo | hi] (pattern 1, the B operand), hi = bf16(x),
 * lo = bf16(x - hi); rows R..rows_out-1 are zero.  A.B^t over K = 3P then equals x.z up to the dropped lo.lo
 * term (~2^-17 relative).  P % 8 == 0. */
int pgca_split_bf16(const float* x, int32_t R, int32_t P, int32_t rows_out, int32_t pattern, void* y_bf16,
                    void* stream);
/* y (+)= alpha * x on flat f32 buffers (gradient un-scaling after all-reduce etc.). */
int pgca_axpy(const float* x, float alpha, float* y, int64_t n, int32_t accumulate, void* stream);
/* gather / scatter rows of an f32 or bf16 [*, H] matrix by int32 row_map. */
int pgca_gather_rows_bf16(const void* src, const int32_t* row_map, int32_t M, int32_t H, void* dst, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PGCA_HIP_H */
