// LayerNorm forward/backward, decoder/text embedding (+LN) forward/backward, column sums.
// HBM-bound row kernels: one 64-lane wave per row, 16-B loads, statistics in registers
// (two-pass mean / centred variance like F.layer_norm), gamma/beta gradients reduced
// per block in registers -> LDS -> one partial row per block (no atomics).
#include <string.h>

#include "common.h"

using namespace pgca;

namespace {

constexpr int MAXV = 8;  // float4 per lane -> H <= 2048

struct RowVec {
  float4 v[MAXV];
};

template <int NV>
__device__ __forceinline__ void load_row_f32(const float* __restrict__ p, int H, int lane, float4 (&v)[NV]) {
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c = (j * 64 + lane) * 4;
    v[j] = c < H ? *reinterpret_cast<const float4*>(p + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
template <int NV>
__device__ __forceinline__ void load_row_bf16(const bf16_t* __restrict__ p, int H, int lane, float4 (&v)[NV]) {
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c = (j * 64 + lane) * 4;
    if (c < H) {
      bf16x4 t = *reinterpret_cast<const bf16x4*>(p + c);
      v[j] = make_float4((float)t[0], (float)t[1], (float)t[2], (float)t[3]);
    } else {
      v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}
__device__ __forceinline__ void store_bf16x4(bf16_t* p, float4 v) {
  bf16x4 t;
  t[0] = (bf16_t)v.x;
  t[1] = (bf16_t)v.y;
  t[2] = (bf16_t)v.z;
  t[3] = (bf16_t)v.w;
  *reinterpret_cast<bf16x4*>(p) = t;
}

template <int NV>
__device__ __forceinline__ void row_stats(const float4 (&v)[NV], int H, int lane, float eps, float& mean,
                                          float& rstd) {
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) s += v[j].x + v[j].y + v[j].z + v[j].w;
  mean = wave_sum(s) / (float)H;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c = (j * 64 + lane) * 4;
    if (c < H) {
      const float a = v[j].x - mean, b = v[j].y - mean, cc = v[j].z - mean, d = v[j].w - mean;
      q += a * a + b * b + cc * cc + d * d;
    }
  }
  rstd = rsqrtf(wave_sum(q) / (float)H + eps);
}

// ------------------------------------------------------------------------------------ LN forward
template <int NV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const int* __restrict__ row_map,
                                                     int M, int H, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float eps,
                                                     bf16_t* __restrict__ yb, float* __restrict__ yf,
                                                     float* __restrict__ mean_o, float* __restrict__ rstd_o) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const size_t src = row_map ? (size_t)row_map[m] : (size_t)m;
  float4 v[NV];
  load_row_f32<NV>(x + src * H, H, lane, v);
  float mean, rstd;
  row_stats<NV>(v, H, lane, eps, mean, rstd);
  if (lane == 0) {
    if (mean_o) mean_o[m] = mean;
    if (rstd_o) rstd_o[m] = rstd;
  }
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c = (j * 64 + lane) * 4;
    if (c < H) {
      const float4 g = *reinterpret_cast<const float4*>(gamma + c);
      const float4 b = *reinterpret_cast<const float4*>(beta + c);
      float4 y;
      y.x = (v[j].x - mean) * rstd * g.x + b.x;
      y.y = (v[j].y - mean) * rstd * g.y + b.y;
      y.z = (v[j].z - mean) * rstd * g.z + b.z;
      y.w = (v[j].w - mean) * rstd * g.w + b.w;
      if (yf) *reinterpret_cast<float4*>(yf + (size_t)m * H + c) = y;
      if (yb) store_bf16x4(yb + (size_t)m * H + c, y);
    }
  }
}

// ------------------------------------------------------------------------------------ LN backward
// Shared tail: reduce the per-lane dgamma/dbeta accumulators of the block's 4 waves through LDS and
// write one partial row per block: part[0][blk][H] (dgamma), part[1][blk][H] (dbeta).
template <int NV>
__device__ __forceinline__ void write_partials(float4 (&dg)[NV], float4 (&db)[NV], int H, float* __restrict__ part,
                                               int nblk, float* red /* [2][4][NV*256] */) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    *reinterpret_cast<float4*>(red + ((0 * 4 + w) * NV + j) * 256 + lane * 4) = dg[j];
    *reinterpret_cast<float4*>(red + ((1 * 4 + w) * NV + j) * 256 + lane * 4) = db[j];
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < 2 * NV * 256; idx += 256) {
    const int which = idx / (NV * 256), c = idx % (NV * 256);
    if (c < H) {
      float s = 0.f;
#pragma unroll
      for (int ww = 0; ww < 4; ++ww) s += red[(which * 4 + ww) * NV * 256 + c];
      part[((size_t)which * nblk + blockIdx.x) * H + c] = s;
    }
  }
}

template <int NV, bool EXTRA>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16_t* __restrict__ dyb, const float* __restrict__ dyf,
                                                     const float* __restrict__ x, const int* __restrict__ row_map,
                                                     int M, int H, const float* __restrict__ gamma,
                                                     const float* __restrict__ mean_i, const float* __restrict__ rstd_i,
                                                     const float* __restrict__ add_to, float* __restrict__ dx_out,
                                                     bf16_t* __restrict__ dx_bf, float* __restrict__ part,
                                                     float* __restrict__ part_extra, Drop drop_add, Drop drop_dx,
                                                     const int* __restrict__ drop_rows) {
  extern __shared__ __attribute__((aligned(16))) float red[];
  const int lane = threadIdx.x & 63;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nw = gridDim.x * 4;
  float4 dg[NV], db[NV], gm[NV], sa[NV], sd[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) sa[j] = sd[j] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    dg[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    db[j] = dg[j];
    const int c = (j * 64 + lane) * 4;
    gm[j] = c < H ? *reinterpret_cast<const float4*>(gamma + c) : dg[j];
  }
  for (int m = wid; m < M; m += nw) {
    const size_t row = row_map ? (size_t)row_map[m] : (size_t)m;
    float4 xv[NV], dy[NV];
    load_row_f32<NV>(x + row * H, H, lane, xv);
    if (dyb)
      load_row_bf16<NV>(dyb + (size_t)m * H, H, lane, dy);
    else
      load_row_f32<NV>(dyf + (size_t)m * H, H, lane, dy);
    // EXTRA (168 VGPRs, 3 waves/SIMD either way): the residual-stream gradient that joins here is requested with the row
    // itself - one HBM round trip per row, not two; it is only consumed after the two wave reductions below
    // (152 -> 128 us at 32768 x 1024).  Without EXTRA the 16 extra registers would cost a wave per SIMD (measured
    // 95 -> 120 us), so that variant keeps the late load.
    float4 av[EXTRA ? NV : 1];
    if constexpr (EXTRA) {
      if (add_to) load_row_f32<NV>(add_to + row * H, H, lane, av);
    }
    const float mean = mean_i[m], rstd = rstd_i[m];
    // packed token rows: the dropout hashes stay keyed on the token's position in the padded batch
    const unsigned drow = drop_rows ? (unsigned)drop_rows[row] : (unsigned)row;
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int c = (j * 64 + lane) * 4;
      if (c < H) {
        xv[j].x = (xv[j].x - mean) * rstd;
        xv[j].y = (xv[j].y - mean) * rstd;
        xv[j].z = (xv[j].z - mean) * rstd;
        xv[j].w = (xv[j].w - mean) * rstd;
        dg[j].x += dy[j].x * xv[j].x; dg[j].y += dy[j].y * xv[j].y;
        dg[j].z += dy[j].z * xv[j].z; dg[j].w += dy[j].w * xv[j].w;
        db[j].x += dy[j].x; db[j].y += dy[j].y; db[j].z += dy[j].z; db[j].w += dy[j].w;
        dy[j].x *= gm[j].x; dy[j].y *= gm[j].y; dy[j].z *= gm[j].z; dy[j].w *= gm[j].w;
        c1 += dy[j].x + dy[j].y + dy[j].z + dy[j].w;
        c2 += dy[j].x * xv[j].x + dy[j].y * xv[j].y + dy[j].z * xv[j].z + dy[j].w * xv[j].w;
      }
    }
    c1 = wave_sum(c1) / (float)H;
    c2 = wave_sum(c2) / (float)H;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int c = (j * 64 + lane) * 4;
      if (c < H) {
        float4 d;
        d.x = rstd * (dy[j].x - c1 - xv[j].x * c2);
        d.y = rstd * (dy[j].y - c1 - xv[j].y * c2);
        d.z = rstd * (dy[j].z - c1 - xv[j].z * c2);
        d.w = rstd * (dy[j].w - c1 - xv[j].w * c2);
        const unsigned eidx = drow * (unsigned)H + (unsigned)c;
        if (add_to) {
          float4 a = EXTRA ? av[EXTRA ? j : 0] : *reinterpret_cast<const float4*>(add_to + row * H + c);
          d.x += a.x; d.y += a.y; d.z += a.z; d.w += a.w;
          if (EXTRA) {
            if (drop_add.on()) {  // bias gradient of the GEMM whose dropped output joined this stream
              float m[4];
              drop_add.mul4(eidx, m);
              a.x *= m[0]; a.y *= m[1]; a.z *= m[2]; a.w *= m[3];
            }
            sa[j].x += a.x; sa[j].y += a.y; sa[j].z += a.z; sa[j].w += a.w;
          }
        }
        *reinterpret_cast<float4*>(dx_out + row * H + c) = d;
        if (drop_dx.on()) {  // the bf16 copy feeds the backward of a GEMM whose output was dropped: replay its mask
          float m[4];
          drop_dx.mul4(eidx, m);
          d.x *= m[0]; d.y *= m[1]; d.z *= m[2]; d.w *= m[3];
        }
        if (EXTRA) { sd[j].x += d.x; sd[j].y += d.y; sd[j].z += d.z; sd[j].w += d.w; }
        if (dx_bf) store_bf16x4(dx_bf + row * H + c, d);
      }
    }
  }
  if (part) write_partials<NV>(dg, db, H, part, gridDim.x, red);
  if (EXTRA) {
    // column sums of the two residual-stream gradients this pass touches anyway: sum(add_to) is the bias
    // gradient of the GEMM that produced this block's output, sum(dx_out) that of the one feeding this LN's input
    __syncthreads();
    write_partials<NV>(sa, sd, H, part_extra, gridDim.x, red);
  }
}

// 1-key cross-attention with attention-weight dropout (nn.MultiheadAttention(dropout=p), reference model.py:528-533):
// softmax over one key is 1, dropout turns it into w(b,h,s) in {0, 1/(1-p)} per head, so
//   attended[b,s,:] = b_o + sum_h w(b,h,s) * U[b,h,:],   U[b,h,:] = W_o[:, head h] . v_b[head h].
// Without dropout every w is 1 and the sum collapses to one vector per b (the `att` path).
struct XAttn {
  const float* U;  // [B, heads, H] or NULL
  float* dU;       // backward only
  int heads;
  Drop drop;
  __device__ __forceinline__ float w(int b, int h, int s, int S) const {
    return drop.on() ? drop.mul(((unsigned)b * heads + h) * (unsigned)S + s) : 1.f;
  }
};

template <int NV>
__device__ __forceinline__ void add_head_terms(const XAttn& xa, int b, int s, int S, int H, int lane, float4 (&v)[NV]) {
  if (!xa.U) return;
  for (int h = 0; h < xa.heads; ++h) {
    const float w = xa.w(b, h, s, S);
    if (w != 0.f) {
      float4 t[NV];
      load_row_f32<NV>(xa.U + ((size_t)b * xa.heads + h) * H, H, lane, t);
#pragma unroll
      for (int j = 0; j < NV; ++j) { v[j].x += w * t[j].x; v[j].y += w * t[j].y; v[j].z += w * t[j].z; v[j].w += w * t[j].w; }
    }
  }
}

// ------------------------------------------------------------------------------------ embeddings
template <int NV>
__global__ __launch_bounds__(256) void embed_fwd_kernel(const long long* __restrict__ ids, int B, int S, int H,
                                                        const float* __restrict__ wte, const float* __restrict__ wpe,
                                                        const float* __restrict__ att, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps,
                                                        float* __restrict__ h0, float* __restrict__ mean_o,
                                                        float* __restrict__ rstd_o, int att_stride, XAttn xa,
                                                        Drop drop_e, const int* __restrict__ row_ids, int n_rows) {
  const int lane = threadIdx.x & 63;
  const int orow = blockIdx.x * 4 + (threadIdx.x >> 6);  // output row; m = the position b*S + s it holds
  if (orow >= n_rows) return;
  const int m = row_ids ? row_ids[orow] : orow;
  if (m < 0) {  // filler row of the packed layout: zeros (finite through every later kernel, zero gradient)
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int c = (j * 64 + lane) * 4;
      if (c < H) *reinterpret_cast<float4*>(h0 + (size_t)orow * H + c) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (gamma && lane == 0) { mean_o[orow] = 0.f; rstd_o[orow] = 1.f; }
    return;
  }
  const int b = m / S, s = m % S;
  float4 v[NV], t[NV];
  load_row_f32<NV>(wte + (size_t)ids[m] * H, H, lane, v);
  if (att) {
    load_row_f32<NV>(att + (size_t)b * att_stride, H, lane, t);
#pragma unroll
    for (int j = 0; j < NV; ++j) { v[j].x += t[j].x; v[j].y += t[j].y; v[j].z += t[j].z; v[j].w += t[j].w; }
  }
  add_head_terms<NV>(xa, b, s, S, H, lane, v);
  float mean = 0.f, rstd = 1.f;
  if (gamma) {
    row_stats<NV>(v, H, lane, eps, mean, rstd);
    if (lane == 0) { mean_o[orow] = mean; rstd_o[orow] = rstd; }
  }
  load_row_f32<NV>(wpe + (size_t)s * H, H, lane, t);
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c = (j * 64 + lane) * 4;
    if (c < H) {
      float4 y = v[j];
      if (gamma) {
        const float4 g = *reinterpret_cast<const float4*>(gamma + c);
        const float4 bb = *reinterpret_cast<const float4*>(beta + c);
        y.x = (y.x - mean) * rstd * g.x + bb.x; y.y = (y.y - mean) * rstd * g.y + bb.y;
        y.z = (y.z - mean) * rstd * g.z + bb.z; y.w = (y.w - mean) * rstd * g.w + bb.w;
      }
      y.x += t[j].x; y.y += t[j].y; y.z += t[j].z; y.w += t[j].w;
      if (drop_e.on()) {  // GPT-2 embedding dropout (modeling_gpt2.py:604)
        const unsigned e = (unsigned)m * (unsigned)H + (unsigned)c;
        float m4[4];
        drop_e.mul4(e, m4);
        y.x *= m4[0]; y.y *= m4[1]; y.z *= m4[2]; y.w *= m4[3];
      }
      *reinterpret_cast<float4*>(h0 + (size_t)orow * H + c) = y;
    }
  }
}

__device__ __forceinline__ void atomic_add4(float* p, float4 d) {
  atomicAdd(p + 0, d.x);
  atomicAdd(p + 1, d.y);
  atomicAdd(p + 2, d.z);
  atomicAdd(p + 3, d.w);
}

// One wave owns `chunk` consecutive positions of ONE sequence (chunk divides S), so the gradients that are shared
// by all positions of a sequence - the attended row and the per-head U rows of the cross-attention - are summed
// in registers and leave the wave as one atomic per chunk instead of one per row (the 128-way contended atomics
// were 2.6 ms of the step).  The position-embedding gradient (shared ACROSS sequences) is wpe_grad_kernel's.
// ACCU: keep the 8 per-head sums in registers (H <= 1024); wider rows fall back to per-row atomics for dU.
template <int NV, bool ACCU>
__global__ __launch_bounds__(256) void embed_bwd_kernel(const float* __restrict__ g, const long long* __restrict__ ids,
                                                        const int* __restrict__ row_mask, int B, int S, int H,
                                                        const float* __restrict__ wte, const float* __restrict__ att,
                                                        const float* __restrict__ gamma, const float* __restrict__ mean_i,
                                                        const float* __restrict__ rstd_i, float* __restrict__ dwte,
                                                        float* __restrict__ datt, float* __restrict__ part,
                                                        int att_stride, XAttn xa, Drop drop_e, int chunk,
                                                        const int* __restrict__ cu) {
  extern __shared__ __attribute__((aligned(16))) float red[];
  const int lane = threadIdx.x & 63;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nw = gridDim.x * 4;
  constexpr int NH = ACCU ? 8 : 1;
  float4 dg[NV], db[NV], gm[NV];
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    dg[j] = zero;
    db[j] = zero;
    const int c = (j * 64 + lane) * 4;
    gm[j] = (gamma && c < H) ? *reinterpret_cast<const float4*>(gamma + c) : zero;
  }
  const int nchunks = (B * S) / chunk;
  for (int ch = wid; ch < nchunks; ch += nw) {
    const int mbase = ch * chunk;
    const int b = mbase / S;
    // packed rows: position (b, s) of the batch lives at row cu[b] + s of g / mean / rstd, for s < len
    const int prow0 = cu ? cu[b] : b * S;
    const int plen = cu ? cu[b + 1] - prow0 : S;
    float4 sa[NV], su[NH][NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      sa[j] = zero;
#pragma unroll
      for (int h = 0; h < NH; ++h) su[h][j] = zero;
    }
    bool any = false;
    for (int r = 0; r < chunk; ++r) {
      const int m = mbase + r;
      if (row_mask && row_mask[m] == 0) continue;  // gradient of a padded position is exactly zero
      const int s = m - b * S;
      if (s >= plen) continue;
      any = true;
      const size_t pr = (size_t)(prow0 + s);
      const long long id = ids[m];
      float4 dy[NV];
      load_row_f32<NV>(g + pr * H, H, lane, dy);
      if (drop_e.on()) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
          const unsigned e = (unsigned)m * (unsigned)H + (unsigned)((j * 64 + lane) * 4);
          float m4[4];
          drop_e.mul4(e, m4);
          dy[j].x *= m4[0]; dy[j].y *= m4[1]; dy[j].z *= m4[2]; dy[j].w *= m4[3];
        }
      }
      if (gamma) {
        float4 xv[NV], t[NV];
        load_row_f32<NV>(wte + (size_t)id * H, H, lane, xv);
        if (att) {
          load_row_f32<NV>(att + (size_t)b * att_stride, H, lane, t);
#pragma unroll
          for (int j = 0; j < NV; ++j) { xv[j].x += t[j].x; xv[j].y += t[j].y; xv[j].z += t[j].z; xv[j].w += t[j].w; }
        }
        add_head_terms<NV>(xa, b, s, S, H, lane, xv);
        const float mean = mean_i[pr], rstd = rstd_i[pr];
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
          const int c = (j * 64 + lane) * 4;
          if (c < H) {
            xv[j].x = (xv[j].x - mean) * rstd; xv[j].y = (xv[j].y - mean) * rstd;
            xv[j].z = (xv[j].z - mean) * rstd; xv[j].w = (xv[j].w - mean) * rstd;
            dg[j].x += dy[j].x * xv[j].x; dg[j].y += dy[j].y * xv[j].y;
            dg[j].z += dy[j].z * xv[j].z; dg[j].w += dy[j].w * xv[j].w;
            db[j].x += dy[j].x; db[j].y += dy[j].y; db[j].z += dy[j].z; db[j].w += dy[j].w;
            dy[j].x *= gm[j].x; dy[j].y *= gm[j].y; dy[j].z *= gm[j].z; dy[j].w *= gm[j].w;
            c1 += dy[j].x + dy[j].y + dy[j].z + dy[j].w;
            c2 += dy[j].x * xv[j].x + dy[j].y * xv[j].y + dy[j].z * xv[j].z + dy[j].w * xv[j].w;
          }
        }
        c1 = wave_sum(c1) / (float)H;
        c2 = wave_sum(c2) / (float)H;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
          dy[j].x = rstd * (dy[j].x - c1 - xv[j].x * c2); dy[j].y = rstd * (dy[j].y - c1 - xv[j].y * c2);
          dy[j].z = rstd * (dy[j].z - c1 - xv[j].z * c2); dy[j].w = rstd * (dy[j].w - c1 - xv[j].w * c2);
        }
      }
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int c = (j * 64 + lane) * 4;
        if (c < H) {
          atomic_add4(dwte + (size_t)id * H + c, dy[j]);
          sa[j].x += dy[j].x; sa[j].y += dy[j].y; sa[j].z += dy[j].z; sa[j].w += dy[j].w;
        }
      }
      if (xa.dU) {
        if (ACCU) {
#pragma unroll
          for (int h = 0; h < NH; ++h) {
            if (h < xa.heads) {
              const float w = xa.w(b, h, s, S);
#pragma unroll
              for (int j = 0; j < NV; ++j) {
                su[h][j].x += dy[j].x * w; su[h][j].y += dy[j].y * w; su[h][j].z += dy[j].z * w; su[h][j].w += dy[j].w * w;
              }
            }
          }
        } else {
          for (int h = 0; h < xa.heads; ++h) {
            const float w = xa.w(b, h, s, S);
            if (w != 0.f) {
#pragma unroll
              for (int j = 0; j < NV; ++j) {
                const int c = (j * 64 + lane) * 4;
                if (c < H)
                  atomic_add4(xa.dU + ((size_t)b * xa.heads + h) * H + c,
                              make_float4(dy[j].x * w, dy[j].y * w, dy[j].z * w, dy[j].w * w));
              }
            }
          }
        }
      }
    }
    if (any) {
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int c = (j * 64 + lane) * 4;
        if (c < H) {
          if (datt) atomic_add4(datt + (size_t)b * H + c, sa[j]);
          if (ACCU && xa.dU) {
#pragma unroll
            for (int h = 0; h < NH; ++h)
              if (h < xa.heads) atomic_add4(xa.dU + ((size_t)b * xa.heads + h) * H + c, su[h][j]);
          }
        }
      }
    }
  }
  if (part && gamma) write_partials<NV>(dg, db, H, part, gridDim.x, red);
}

// dwpe[s, :] += sum over sequences b of g[b, s, :] (masked rows skipped, embedding dropout replayed).
// One thread owns 4 columns of one position and walks a slice of the batch: every load is a coalesced 16-B
// vector, the gridDim.y batch slices meet through (gridDim.y-way) atomics.
__global__ __launch_bounds__(256) void wpe_grad_kernel(const float* __restrict__ g, const int* __restrict__ row_mask,
                                                       int B, int S, int H, float* __restrict__ dwpe, Drop drop_e,
                                                       int bslice, const int* __restrict__ cu) {
  const int h4 = H >> 2;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= S * h4) return;
  const int s = idx / h4, c = (idx - s * h4) * 4;
  const int b0 = blockIdx.y * bslice, b1 = min(B, b0 + bslice);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
  for (int b = b0; b < b1; ++b) {
    const int m = b * S + s;
    if (row_mask && row_mask[m] == 0) continue;
    size_t pr = (size_t)m;
    if (cu) {
      const int p0 = cu[b];
      if (s >= cu[b + 1] - p0) continue;
      pr = (size_t)(p0 + s);
    }
    float4 v = *reinterpret_cast<const float4*>(g + pr * H + c);
    if (drop_e.on()) {
      const unsigned e = (unsigned)m * (unsigned)H + (unsigned)c;
      float m4[4];
      drop_e.mul4(e, m4);
      v.x *= m4[0]; v.y *= m4[1]; v.z *= m4[2]; v.w *= m4[3];
    }
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  atomic_add4(dwpe + (size_t)s * H + c, acc);
}

// ------------------------------------------------------------------------------------ column sums
// out[c] (+)= sum_b part[b, c].  Block = 32 columns x 8 row slices: every load is 128 contiguous bytes,
// each thread sums nparts/8 independent partials, the 8 slices meet in LDS.
__global__ __launch_bounds__(256) void colsum_finish_kernel(const float* __restrict__ part, int nparts, int H,
                                                            float* __restrict__ out, int accumulate) {
  __shared__ float red[8][32];
  const int cx = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cx;
  float s = 0.f;
  if (c < H) {
#pragma unroll 4
    for (int b = sl; b < nparts; b += 8) s += part[(size_t)b * H + c];
  }
  red[sl][cx] = s;
  __syncthreads();
  if (sl == 0 && c < H) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red[k][cx];
    out[c] = accumulate ? out[c] + t : t;
  }
}

// Up to four planes part[plane][b][c] folded into four outputs in one launch (blockIdx.y = plane).
__global__ __launch_bounds__(256) void colsum_finish4_kernel(const float* __restrict__ part, int nparts, int H,
                                                             float* o0, float* o1, float* o2, float* o3,
                                                             int accumulate) {
  __shared__ float red[8][32];
  const int cx = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cx;
  const float* pl = part + (size_t)blockIdx.y * nparts * H;
  float* out = blockIdx.y == 0 ? o0 : (blockIdx.y == 1 ? o1 : (blockIdx.y == 2 ? o2 : o3));
  float s = 0.f;
  if (c < H) {
#pragma unroll 4
    for (int b = sl; b < nparts; b += 8) s += pl[(size_t)b * H + c];
  }
  red[sl][cx] = s;
  __syncthreads();
  if (sl == 0 && c < H) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red[k][cx];
    out[c] = accumulate ? out[c] + t : t;
  }
}

// part[blockIdx.y, n] = sum over this block's rows of x[m, n].  Block = 32 column groups (8 columns = one 16-B
// bf16 load) x 8 row lanes; a block covers 256 columns x rows_per_blk rows; row lanes meet in LDS.
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ xb, const float* __restrict__ xf,
                                                     int M, int N, int ld, int rows_per_blk, float* __restrict__ part) {
  __shared__ float red[8][32][9];
  const int cg = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int c0 = (blockIdx.x * 32 + cg) * 8;
  const int r0 = blockIdx.y * rows_per_blk;
  const int r1 = min(M, r0 + rows_per_blk);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c0 < N) {
    const bool full = c0 + 8 <= N;
    for (int r = r0 + ry; r < r1; r += 8) {
      if (xb) {
        if (full) {
          const bf16x8 v = *reinterpret_cast<const bf16x8*>(xb + (size_t)r * ld + c0);
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[i] += (float)v[i];
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) if (c0 + i < N) acc[i] += (float)xb[(size_t)r * ld + c0 + i];
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) if (c0 + i < N) acc[i] += xf[(size_t)r * ld + c0 + i];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) red[ry][cg][i] = acc[i];
  __syncthreads();
  // 256 threads -> 256 columns of this block
  const int col = threadIdx.x;
  const int gcol = blockIdx.x * 256 + col;
  if (gcol < N) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += red[k][col >> 3][col & 7];
    part[(size_t)blockIdx.y * N + gcol] = s;
  }
}

int nv_for(int H) { return (H + 255) / 256; }

}  // namespace

#define DISPATCH_NV(NVV, CALL)                       \
  switch (NVV) {                                     \
    case 1: { constexpr int NV = 1; CALL; } break;   \
    case 2: { constexpr int NV = 2; CALL; } break;   \
    case 3: { constexpr int NV = 3; CALL; } break;   \
    case 4: { constexpr int NV = 4; CALL; } break;   \
    case 5: { constexpr int NV = 5; CALL; } break;   \
    case 6: { constexpr int NV = 6; CALL; } break;   \
    case 7: { constexpr int NV = 7; CALL; } break;   \
    default: { constexpr int NV = 8; CALL; } break;  \
  }

static int check_h(const char* who, int H) {
  if (H <= 0 || (H & 3) || H > MAXV * 256) {
    set_error("%s: H=%d must be a positive multiple of 4 and <= %d", who, H, MAXV * 256);
    return PGCA_ERR_INVALID;
  }
  return PGCA_OK;
}

extern "C" int pgca_layernorm_fwd(const float* x, const int32_t* row_map, int32_t M, int32_t H, const float* gamma,
                                  const float* beta, float eps, void* y_bf16, float* y_f32, float* mean, float* rstd,
                                  void* stream) {
  if (check_h("pgca_layernorm_fwd", H)) return PGCA_ERR_INVALID;
  if (!x || !gamma || !beta || M <= 0 || (!y_bf16 && !y_f32)) {
    set_error("pgca_layernorm_fwd: bad arguments");
    return PGCA_ERR_INVALID;
  }
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((M + 3) / 4), block(256);
  DISPATCH_NV(nv_for(H), hipLaunchKernelGGL((ln_fwd_kernel<NV>), grid, block, 0, s, x, row_map, M, H, gamma, beta,
                                            eps, (bf16_t*)y_bf16, y_f32, mean, rstd));
  return check_launch("pgca_layernorm_fwd");
}

extern "C" int pgca_layernorm_bwd_blocks(int32_t M) {
  int b = (M + 3) / 4;
  return b < 1 ? 1 : (b > 1024 ? 1024 : b);
}

extern "C" int pgca_layernorm_bwd(const void* dy_bf16, const float* dy_f32, const float* x, const int32_t* row_map,
                                  int32_t M, int32_t H, const float* gamma, const float* mean, const float* rstd,
                                  const float* add_to, float* dx_out, void* dx_bf16, float* part, float* part_extra,
                                  const uint32_t* drop_add, const uint32_t* drop_dx, const int32_t* drop_rows,
                                  void* stream) {
  if (check_h("pgca_layernorm_bwd", H)) return PGCA_ERR_INVALID;
  if ((!dy_bf16) == (!dy_f32) || !x || !gamma || !mean || !rstd || !dx_out || M <= 0 || (part_extra && !part)) {
    set_error("pgca_layernorm_bwd: bad arguments");
    return PGCA_ERR_INVALID;
  }
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(pgca_layernorm_bwd_blocks(M)), block(256);
  const int nv = nv_for(H);
  const size_t lds = (size_t)2 * 4 * nv * 256 * sizeof(float);
  // drop_* point at {seed, threshold, float bits of scale} (host memory) or are NULL
  auto mk = [](const uint32_t* d) {
    Drop r{0u, 0u, 1.f};
    if (d) {
      r.seed = d[0];
      r.threshold = d[1];
      memcpy(&r.scale, &d[2], sizeof(float));
    }
    return r;
  };
  const Drop da = mk(drop_add), dd = mk(drop_dx);
  if (part_extra) {
    DISPATCH_NV(nv, hipLaunchKernelGGL((ln_bwd_kernel<NV, true>), grid, block, lds, s, (const bf16_t*)dy_bf16, dy_f32,
                                       x, row_map, M, H, gamma, mean, rstd, add_to, dx_out, (bf16_t*)dx_bf16, part,
                                       part_extra, da, dd, drop_rows));
  } else {
    DISPATCH_NV(nv, hipLaunchKernelGGL((ln_bwd_kernel<NV, false>), grid, block, lds, s, (const bf16_t*)dy_bf16, dy_f32,
                                       x, row_map, M, H, gamma, mean, rstd, add_to, dx_out, (bf16_t*)dx_bf16, part,
                                       part_extra, da, dd, drop_rows));
  }
  return check_launch("pgca_layernorm_bwd");
}

extern "C" int pgca_colsum_finish(const float* part, int32_t nparts, int32_t H, float* out, int32_t accumulate,
                                  void* stream) {
  if (!part || !out || nparts <= 0 || H <= 0) {
    set_error("pgca_colsum_finish: bad arguments");
    return PGCA_ERR_INVALID;
  }
  hipLaunchKernelGGL(colsum_finish_kernel, dim3((H + 31) / 32), dim3(256), 0, (hipStream_t)stream, part, nparts, H,
                     out, accumulate);
  return check_launch("pgca_colsum_finish");
}

extern "C" int pgca_colsum_finish4(const float* part, int32_t nplanes, int32_t nparts, int32_t H, float* out0,
                                   float* out1, float* out2, float* out3, int32_t accumulate, void* stream) {
  if (!part || nplanes < 1 || nplanes > 4 || nparts <= 0 || H <= 0 || !out0 || (nplanes > 1 && !out1) ||
      (nplanes > 2 && !out2) || (nplanes > 3 && !out3)) {
    set_error("pgca_colsum_finish4: bad arguments");
    return PGCA_ERR_INVALID;
  }
  hipLaunchKernelGGL(colsum_finish4_kernel, dim3((H + 31) / 32, nplanes), dim3(256), 0, (hipStream_t)stream, part,
                     nparts, H, out0, out1, out2, out3, accumulate);
  return check_launch("pgca_colsum_finish4");
}

extern "C" int pgca_colsum_blocks(int32_t M) {
  int b = (M + 63) / 64;
  return b < 1 ? 1 : (b > 256 ? 256 : b);
}

extern "C" int pgca_colsum(const void* x_bf16, const float* x_f32, int32_t M, int32_t N, int32_t ld, float* part,
                           void* stream) {
  if ((!x_bf16) == (!x_f32) || !part || M <= 0 || N <= 0 || ld < N || (x_bf16 && (ld & 7))) {
    set_error("pgca_colsum: bad arguments");
    return PGCA_ERR_INVALID;
  }
  const int nb = pgca_colsum_blocks(M);
  const int rows = (M + nb - 1) / nb;
  dim3 grid((N + 255) / 256, nb), block(256);
  hipLaunchKernelGGL(colsum_kernel, grid, block, 0, (hipStream_t)stream, (const bf16_t*)x_bf16, x_f32, M, N, ld, rows,
                     part);
  return check_launch("pgca_colsum");
}

static Drop drop_from_words(const uint32_t* d) {
  Drop r{0u, 0u, 1.f};
  if (d) {
    r.seed = d[0];
    r.threshold = d[1];
    memcpy(&r.scale, &d[2], sizeof(float));
  }
  return r;
}

extern "C" int pgca_embed_fwd(const int64_t* ids, int32_t B, int32_t S, int32_t H, const float* wte, const float* wpe,
                              const float* attended, const float* gamma, const float* beta, float eps, float* h0,
                              float* mean, float* rstd, int32_t att_stride, const float* U, int32_t xheads,
                              const uint32_t* drop_x, const uint32_t* drop_e, const int32_t* row_ids, int32_t n_rows,
                              void* stream) {
  if (check_h("pgca_embed_fwd", H)) return PGCA_ERR_INVALID;
  if (!ids || !wte || !wpe || !h0 || B <= 0 || S <= 0 || (gamma && (!beta || !mean || !rstd)) ||
      (row_ids && n_rows <= 0)) {
    set_error("pgca_embed_fwd: bad arguments");
    return PGCA_ERR_INVALID;
  }
  hipStream_t s = (hipStream_t)stream;
  if (!row_ids) n_rows = B * S;
  dim3 grid((n_rows + 3) / 4), block(256);
  const XAttn xa{U, nullptr, xheads, drop_from_words(drop_x)};
  const Drop de = drop_from_words(drop_e);
  DISPATCH_NV(nv_for(H), hipLaunchKernelGGL((embed_fwd_kernel<NV>), grid, block, 0, s, (const long long*)ids, B, S, H,
                                            wte, wpe, attended, gamma, beta, eps, h0, mean, rstd, att_stride, xa, de,
                                            row_ids, n_rows));
  return check_launch("pgca_embed_fwd");
}

static int embed_chunk(int S) {
  int c = 16;
  while (c > 1 && (S % c)) c >>= 1;
  return c;
}

extern "C" int pgca_embed_bwd_blocks(int32_t B, int32_t S) {
  const int nchunks = (B * S) / embed_chunk(S);
  int b = (nchunks + 3) / 4;
  return b < 1 ? 1 : (b > 1024 ? 1024 : b);
}

extern "C" int pgca_embed_bwd(const float* g, const int64_t* ids, const int32_t* row_mask, int32_t B, int32_t S,
                              int32_t H, const float* wte, const float* attended, const float* gamma,
                              const float* mean, const float* rstd, float* dwte, float* dwpe, float* dattended,
                              float* part, int32_t att_stride, const float* U, float* dU, int32_t xheads,
                              const uint32_t* drop_x, const uint32_t* drop_e, const int32_t* cu_seqlens,
                              void* stream) {
  if (check_h("pgca_embed_bwd", H)) return PGCA_ERR_INVALID;
  if (!g || !ids || !dwte || !dwpe || B <= 0 || S <= 0 || (gamma && (!wte || !mean || !rstd || !part))) {
    set_error("pgca_embed_bwd: bad arguments");
    return PGCA_ERR_INVALID;
  }
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(pgca_embed_bwd_blocks(B, S)), block(256);
  const int nv = nv_for(H);
  const size_t lds = (size_t)2 * 4 * nv * 256 * sizeof(float);
  const XAttn xa{U, dU, xheads, drop_from_words(drop_x)};
  const Drop de = drop_from_words(drop_e);
  const int chunk = embed_chunk(S);
  if (nv <= 4 && xheads <= 8) {
    DISPATCH_NV(nv, hipLaunchKernelGGL((embed_bwd_kernel<(NV <= 4 ? NV : 4), true>), grid, block, lds, s, g,
                                       (const long long*)ids, row_mask, B, S, H, wte, attended, gamma, mean, rstd, dwte,
                                       dattended, part, att_stride, xa, de, chunk, cu_seqlens));
  } else {
    DISPATCH_NV(nv, hipLaunchKernelGGL((embed_bwd_kernel<NV, false>), grid, block, lds, s, g, (const long long*)ids,
                                       row_mask, B, S, H, wte, attended, gamma, mean, rstd, dwte, dattended, part,
                                       att_stride, xa, de, chunk, cu_seqlens));
  }
  {
    const int nthr = S * (H >> 2);
    int ys = B >= 64 ? 8 : (B >= 8 ? 2 : 1);
    const int bslice = (B + ys - 1) / ys;
    ys = (B + bslice - 1) / bslice;
    hipLaunchKernelGGL(wpe_grad_kernel, dim3((nthr + 255) / 256, ys), dim3(256), 0, s, g, row_mask, B, S, H, dwpe, de,
                       bslice, cu_seqlens);
  }
  return check_launch("pgca_embed_bwd");
}
