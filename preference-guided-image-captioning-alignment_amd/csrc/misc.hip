// Small HBM-bound kernels of the step: ViT patch gather, sequence reduce + DPO loss,
// masked mean pool, L2 normalise, NT-Xent scalar reduce, fused clip/AdamW over flat buffers.
#include "common.h"

using namespace pgca;

namespace {

// ------------------------------------------------------------------------------------ ViT input
// out[(b*G*G + gy*G + gx), c*P*P + ky*P + kx] = pixels[b, c, gy*P+ky, gx*P+kx]   (bf16)
__global__ void patchify_kernel(const float* __restrict__ px, int B, int I, int P, bf16_t* __restrict__ out) {
  const int G = I / P, D = 3 * P * P;
  const size_t total = (size_t)B * G * G * D / 4;  // 4 consecutive kx per thread (P % 4 == 0)
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t e = i * 4;
    const int col = (int)(e % D);
    const size_t row = e / D;
    const int c = col / (P * P), ky = (col / P) % P, kx = col % P;
    const int b = (int)(row / (G * G)), gy = (int)(row / G) % G, gx = (int)(row % G);
    const float4 v = *reinterpret_cast<const float4*>(px + (((size_t)b * 3 + c) * I + gy * P + ky) * I + gx * P + kx);
    bf16x4 t;
    t[0] = (bf16_t)v.x; t[1] = (bf16_t)v.y; t[2] = (bf16_t)v.z; t[3] = (bf16_t)v.w;
    *reinterpret_cast<bf16x4*>(out + e) = t;
  }
}

// General patch size (CLIP ViT-L/14: P = 14, D = 588): two consecutive kx per thread (P even), row stride ld >= D,
// columns D..ld-1 written as zeros (K padding of the patch-embedding GEMM to a multiple of 64).
__global__ void patchify_pad_kernel(const float* __restrict__ px, int B, int I, int P, int ld,
                                    bf16_t* __restrict__ out) {
  const int G = I / P, D = 3 * P * P;
  const size_t total = (size_t)B * G * G * (ld / 2);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int col = (int)(i % (ld / 2)) * 2;
    const size_t row = i / (ld / 2);
    bf16x2 t;
    t[0] = t[1] = (bf16_t)0.f;
    if (col < D) {
      const int c = col / (P * P), ky = (col / P) % P, kx = col % P;
      const int b = (int)(row / (G * G)), gy = (int)(row / G) % G, gx = (int)(row % G);
      const float2 v = *reinterpret_cast<const float2*>(px + (((size_t)b * 3 + c) * I + gy * P + ky) * I + gx * P + kx);
      t[0] = (bf16_t)v.x;
      t[1] = (bf16_t)v.y;
    }
    *reinterpret_cast<bf16x2*>(out + row * ld + col) = t;
  }
}

__global__ void vit_assemble_kernel(const float* __restrict__ pe, const float* __restrict__ cls,
                                    const float* __restrict__ pos, int B, int T, int H, float* __restrict__ x) {
  const size_t total = (size_t)B * T * H;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % H);
    const int tkn = (int)((i / H) % T);
    const size_t b = i / ((size_t)H * T);
    const float v = tkn == 0 ? cls[c] : pe[(b * (T - 1) + (tkn - 1)) * H + c];
    x[i] = v + pos[(size_t)tkn * H + c];
  }
}

// One thread per (token, column): the batch loop reads dx with stride T*H (coalesced across columns), casts the patch rows
// and sums over the batch for the position / class-token gradients.
__global__ void vit_assemble_bwd_kernel(const float* __restrict__ dx, int B, int T, int H, bf16_t* __restrict__ dpatch,
                                        float* __restrict__ dcls, float* __restrict__ dpos) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= T * H) return;
  const int tkn = i / H, c = i - tkn * H;
  float s = 0.f;
  for (int b = 0; b < B; ++b) {
    const float v = dx[(size_t)b * T * H + i];
    s += v;
    if (tkn > 0) dpatch[((size_t)b * (T - 1) + (tkn - 1)) * H + c] = f2bf(v);
  }
  dpos[i] += s;
  if (tkn == 0) dcls[c] += s;
}

// ------------------------------------------------------------------------------------ sequence reduce / DPO
// One wave per sequence: compact rows of a sequence are contiguous [row_begin[q], row_begin[q+1]).
__global__ void seq_reduce_kernel(const float* __restrict__ tok, const int* __restrict__ seq_of_row, int nrows,
                                  int nseq, const int* __restrict__ seq_count, int mode, float* __restrict__ out) {
  const int q = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (q >= nseq) return;
  float s = 0.f;
  if (seq_count) {
    // rows are sorted by sequence: this one's start is the sum of the counts before it (<= a few thousand integers,
    // 64 lanes wide) - O(nseq + count) per wave instead of a scan over every row of the batch
    int before = 0;
    for (int i = lane; i < q; i += 64) before += seq_count[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o);
    const int n = seq_count[q];
    for (int r = lane; r < n; r += 64) s += tok[before + r];
  } else {
    for (int r = lane; r < nrows; r += 64)
      if (seq_of_row[r] == q) s += tok[r];
  }
  s = wave_sum(s);
  if (lane == 0) out[q] = mode ? s / (float)seq_count[q] : s;
}

__global__ void row_scale_kernel(const float* __restrict__ dseq, const int* __restrict__ seq_of_row,
                                 const int* __restrict__ seq_count, int nrows, int mode, float* __restrict__ rs) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nrows) return;
  const int q = seq_of_row[r];
  const float v = (mode & 1) ? dseq[q] / (float)seq_count[q] : dseq[q];
  rs[r] = (mode & 2) ? -v : v;
}

__device__ __forceinline__ float log_sigmoid(float z) {  // stable: min(z,0) - log1p(exp(-|z|))
  return fminf(z, 0.f) - log1pf(__expf(-fabsf(z)));
}
__device__ __forceinline__ float sigmoidf(float z) { return 1.f / (1.f + __expf(-z)); }

// single block; B pairs.  A non-finite batch loss (a caption with <= 1 real token makes its length-mean 0/0, reference
// model.py:1082-1083) zeroes the gradient seeds: the micro-batch then contributes nothing, which is what the reference's
// "skip this batch" (trainer.py:481-489,606-613: no backward) amounts to - decided on the device, no host sync.
__global__ void dpo_loss_kernel(const float* __restrict__ pw, const float* __restrict__ pl,
                                const float* __restrict__ rw, const float* __restrict__ rl, int B, float beta,
                                float ls, float* __restrict__ loss, float* __restrict__ dpw, float* __restrict__ dpl,
                                float* __restrict__ metrics) {
  __shared__ float red[5][4];
  __shared__ float s_loss;
  float a_loss = 0.f, a_margin = 0.f, a_acc = 0.f, a_w = 0.f, a_l = 0.f;
  const float tgt = 1.f - ls;  // label smoothing (components.py:223-228): BCE with target (1 - ls); ls == 0 -> -logsigmoid(z)
  for (int i = threadIdx.x; i < B; i += blockDim.x) {
    const float pol = pw[i] - pl[i];
    const float ref = rw ? rw[i] - rl[i] : 0.f;
    const float z = beta * (pol - ref);
    a_loss += -(tgt * log_sigmoid(z) + (1.f - tgt) * log_sigmoid(-z));
    a_margin += pol - ref;
    a_acc += pol > ref ? 1.f : 0.f;
    a_w += pw[i];
    a_l += pl[i];
  }
  float v[5] = {a_loss, a_margin, a_acc, a_w, a_l};
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const float s = wave_sum(v[k]);
    if (lane == 0) red[k][w] = s;
  }
  __syncthreads();
  if (threadIdx.x < 5) {
    const float s = (red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3]) / (float)B;
    if (threadIdx.x == 0) {
      loss[0] = s;
      s_loss = s;
    } else if (metrics) {
      metrics[threadIdx.x - 1] = s;
    }
  }
  __syncthreads();
  if (dpw || dpl) {
    const bool live = isfinite(s_loss);
    for (int i = threadIdx.x; i < B; i += blockDim.x) {
      const float pol = pw[i] - pl[i];
      const float ref = rw ? rw[i] - rl[i] : 0.f;
      const float z = beta * (pol - ref);
      const float dz = live ? -(tgt * (1.f - sigmoidf(z)) - (1.f - tgt) * sigmoidf(z)) / (float)B : 0.f;
      if (dpw) dpw[i] = live ? beta * dz : 0.f;
      if (dpl) dpl[i] = live ? -beta * dz : 0.f;
    }
  }
}

// ------------------------------------------------------------------------------------ batch index preparation
// The integer work of the loss (reference model.py:1069-1083 / components.py:340-357: shift, mask product, gather index)
// on the device, so a batch needs no host pass: rows (b, t) with mask[b, t+1] != 0 are kept, sorted by sequence;
// row_map[r] = b*S + t (the hidden-state row whose logits score the token), targets[r] = ids[b, t+1] (bit-exact int64),
// seq_of_row[r] = b, counts[b] = number of kept rows of sequence b, n_rows[0] = their total.
// One wave per sequence; mask32 (optional output) is the int32 0/1 key mask the attention kernels take.
__global__ void seq_counts_kernel(const long long* __restrict__ mask, int Bq, int S, int* __restrict__ counts,
                                  int* __restrict__ mask32) {
  const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (b >= Bq) return;
  int c = 0;
  for (int t = lane; t < S; t += 64) {
    const int m = mask[(size_t)b * S + t] != 0;
    if (mask32) mask32[(size_t)b * S + t] = m;
    if (t >= 1) c += m;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  if (lane == 0) counts[b] = c;
}
__global__ void seq_compact_kernel(const long long* __restrict__ ids, const long long* __restrict__ mask,
                                   const int* __restrict__ counts, int Bq, int S, int* __restrict__ row_map,
                                   long long* __restrict__ targets, int* __restrict__ seq_of_row,
                                   int* __restrict__ n_rows) {
  const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (b >= Bq) return;
  int off = 0;  // exclusive prefix of the counts: Bq <= a few hundred, every wave sums its own
  for (int i = lane; i < b; i += 64) off += counts[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) off += __shfl_xor(off, o);
  if (b == Bq - 1 && lane == 0) n_rows[0] = off + counts[b];
  for (int t0 = 0; t0 < S - 1; t0 += 64) {
    const int t = t0 + lane;
    const bool keep = t < S - 1 && mask[(size_t)b * S + t + 1] != 0;
    const unsigned long long bal = __ballot(keep);
    if (keep) {
      const int r = off + __popcll(bal & ((1ull << lane) - 1ull));
      row_map[r] = b * S + t;
      targets[r] = ids[(size_t)b * S + t + 1];
      seq_of_row[r] = b;
    }
    off += __popcll(bal);
  }
}

// Packed row layout (pgca_seq_pack_prepare): lens[b] = 1 + last position with a non-zero mask (0 for an empty sequence).
__global__ void seq_lens_kernel(const int* __restrict__ mask32, int Bq, int S, int* __restrict__ lens) {
  const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (b >= Bq) return;
  int len = 0;
  for (int t = lane; t < S; t += 64)
    if (mask32[(size_t)b * S + t] != 0) len = t + 1;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) len = max(len, __shfl_xor(len, o));
  if (lane == 0) lens[b] = len;
}
// One wave per sequence (+ one for the filler tail, b == Bq): offsets, row ids, the filler, and the compact rows' map.
__global__ void seq_pack_kernel(const int* __restrict__ lens, int Bq, int S, int pad_to, int* __restrict__ cu,
                                int* __restrict__ row_ids, int* __restrict__ n_packed, int* __restrict__ mask32,
                                const int* __restrict__ counts, int* __restrict__ row_map) {
  const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (b > Bq) return;
  int off = 0, coff = 0;  // exclusive prefixes of the lengths and of the scored-row counts
  for (int i = lane; i < b; i += 64) {
    off += lens[i];
    if (counts) coff += counts[i];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    off += __shfl_xor(off, o);
    coff += __shfl_xor(coff, o);
  }
  if (b == Bq) {  // filler: rows off .. padded-1, as F unmasked pseudo-sequences of at most S rows each
    const int padded = (off + pad_to - 1) / pad_to * pad_to;
    const int F = (pad_to - 1 + S - 1) / S;
    if (lane == 0) {
      for (int f = 0; f <= F; ++f) cu[Bq + f] = min(off + f * S, padded);
      n_packed[0] = off;
      n_packed[1] = padded;
    }
    for (int r = off + lane; r < padded; r += 64) row_ids[r] = -1;
    for (int t = lane; t < F * S; t += 64) mask32[(size_t)Bq * S + t] = 1;
    return;
  }
  if (lane == 0) cu[b] = off;
  const int len = lens[b];
  for (int t = lane; t < len; t += 64) row_ids[off + t] = b * S + t;
  if (row_map) {
    const int n = counts[b];
    for (int r = lane; r < n; r += 64) row_map[coff + r] += off - b * S;
  }
}

// ------------------------------------------------------------------------------------ pooling / normalise
// cu (optional): packed rows - position (b, t) is row cu[b] + t of f / df, for t < cu[b+1] - cu[b] (every position with a
// non-zero mask lies below that length by construction: pgca_seq_pack_prepare)
__global__ void masked_mean_fwd_kernel(const float* __restrict__ f, const int* __restrict__ mask, int S, int H,
                                       float* __restrict__ pooled, const int* __restrict__ cu) {
  const int b = blockIdx.y;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= H) return;
  const size_t r0 = cu ? (size_t)cu[b] : (size_t)b * S;
  const int len = cu ? cu[b + 1] - cu[b] : S;
  float s = 0.f;
  int cnt = 0;
  for (int t = 0; t < S; ++t) {
    const int m = mask[b * S + t];
    cnt += m;
    if (m && t < len) s += f[(r0 + t) * H + c] * (float)m;
  }
  pooled[(size_t)b * H + c] = s / (float)(cnt < 1 ? 1 : cnt);
}
__global__ void masked_mean_bwd_kernel(const float* __restrict__ dp, const int* __restrict__ mask, int S, int H,
                                       float* __restrict__ df, const int* __restrict__ cu) {
  const int b = blockIdx.y;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= H) return;
  const size_t r0 = cu ? (size_t)cu[b] : (size_t)b * S;
  const int len = cu ? cu[b + 1] - cu[b] : S;
  int cnt = 0;
  for (int t = 0; t < S; ++t) cnt += mask[b * S + t];
  const float g = dp[(size_t)b * H + c] / (float)(cnt < 1 ? 1 : cnt);
  for (int t = 0; t < len; ++t) df[(r0 + t) * H + c] = g * (float)mask[b * S + t];
}

// one wave per row
__global__ void l2norm_fwd_kernel(const float* __restrict__ x, int B, int P, float* __restrict__ y,
                                  float* __restrict__ norm) {
  const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= B) return;
  float s = 0.f;
  for (int c = lane; c < P; c += 64) { const float v = x[(size_t)r * P + c]; s += v * v; }
  const float n = fmaxf(sqrtf(wave_sum(s)), 1e-12f);
  for (int c = lane; c < P; c += 64) y[(size_t)r * P + c] = x[(size_t)r * P + c] / n;
  if (lane == 0 && norm) norm[r] = n;
}
__global__ void l2norm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                  const float* __restrict__ norm, int B, int P, float* __restrict__ dx) {
  const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= B) return;
  float s = 0.f;
  for (int c = lane; c < P; c += 64) s += dy[(size_t)r * P + c] * y[(size_t)r * P + c];
  s = wave_sum(s);
  const float inv = 1.f / norm[r];
  for (int c = lane; c < P; c += 64) dx[(size_t)r * P + c] = (dy[(size_t)r * P + c] - y[(size_t)r * P + c] * s) * inv;
}

__global__ void ntxent_loss_kernel(const float* __restrict__ lr, const float* __restrict__ lc,
                                   const float* __restrict__ diag, int n_local, int n_total, float* __restrict__ loss) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n_local; i += blockDim.x) s += (lr[i] - diag[i]) + (lc[i] - diag[i]);
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) loss[0] = (red[0] + red[1] + red[2] + red[3]) / (2.f * (float)n_total);
}

// ------------------------------------------------------------------------------------ optimiser
constexpr int SQ_BLOCK_ELEMS = 256 * 4 * 16;  // 16 float4 per thread

__global__ __launch_bounds__(256) void sqnorm_kernel(const float* __restrict__ g, long long n, float* __restrict__ part) {
  __shared__ float red[4];
  const long long base = (long long)blockIdx.x * SQ_BLOCK_ELEMS;
  float s = 0.f;
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const long long e = base + ((long long)i * 256 + threadIdx.x) * 4;
    if (e + 4 <= n) {
      const float4 v = *reinterpret_cast<const float4*>(g + e);
      s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    } else {
      for (long long k = e; k < n && k < e + 4; ++k) s += g[k] * g[k];
    }
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ void step_control_kernel(const float* __restrict__ part, int nparts, float max_norm, float base_lr,
                                    int warmup, int total_steps, int sched_stride, float beta1, float beta2,
                                    float grad_scale, const float* __restrict__ gate, float* __restrict__ ctrl) {
  __shared__ double red[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < nparts; i += blockDim.x) s += (double)part[i];
  // wave reduce in double via two floats is lossy; use shuffles on the 64-bit value
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double tot = (red[0] + red[1] + red[2] + red[3]) * (double)grad_scale * (double)grad_scale;
    const float norm = (float)sqrt(tot);
    // `gate` (optional): the loss of the micro-batch that closes the accumulation group.  The reference drops the whole
    // group without stepping when THAT loss is non-finite (its zero_grad() is only real on the boundary micro-step,
    // trainer.py:481-489 under accelerate's accumulate()).
    const bool finite = isfinite(norm) && (!gate || isfinite(gate[0]));
    ctrl[0] = norm;
    ctrl[1] = finite ? 1.f : 0.f;
    ctrl[2] = (max_norm > 0.f) ? fminf(1.f, max_norm / (norm + 1e-6f)) : 1.f;
    if (finite) {
      // lr is read from the schedule BEFORE it advances (optimizer.step() then scheduler.step(),
      // reference trainer.py:518-519,626-627); the schedule advances `sched_stride` per optimiser
      // step (= num_processes under Accelerate).
      const float sched = ctrl[7];
      float mult;
      if (sched < (float)warmup) {
        mult = sched / (float)(warmup < 1 ? 1 : warmup);
      } else {
        const float denom = (float)((total_steps - warmup) < 1 ? 1 : (total_steps - warmup));
        const float prog = (sched - (float)warmup) / denom;
        mult = fmaxf(0.f, 0.5f * (1.f + cosf(3.14159265358979323846f * prog)));
      }
      const float step = ctrl[6] + 1.f;
      ctrl[3] = base_lr * mult;
      ctrl[4] = 1.f - powf(beta1, step);
      ctrl[5] = 1.f - powf(beta2, step);
      ctrl[6] = step;
      ctrl[7] = sched + (float)sched_stride;
    }
  }
}

// clip_grad_norm_ on a partially accumulated gradient (the reference clips after EVERY micro-batch, trainer.py:511-515,
// 619-623): coef[0] = min(1, max_norm / (norm + 1e-6)), norm from the sqnorm partials; then scale_dev multiplies.
__global__ void clip_coef_kernel(const float* __restrict__ part, int nparts, float max_norm, float* __restrict__ coef) {
  __shared__ double red[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < nparts; i += blockDim.x) s += (double)part[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float norm = (float)sqrt(red[0] + red[1] + red[2] + red[3]);
    coef[0] = isfinite(norm) ? fminf(1.f, max_norm / (norm + 1e-6f)) : 1.f;  // NaN gradients are left for the step's check
    coef[1] = norm;
  }
}
__global__ __launch_bounds__(256) void scale_dev_kernel(float* __restrict__ x, long long n, const float* __restrict__ coef) {
  const float c = coef[0];
  if (c == 1.f) return;
  const long long stride = (long long)gridDim.x * blockDim.x * 4;
  for (long long e = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; e < n; e += stride) {
    float4 v = *reinterpret_cast<float4*>(x + e);
    v.x *= c; v.y *= c; v.z *= c; v.w *= c;
    *reinterpret_cast<float4*>(x + e) = v;
  }
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v,
                                                    bf16_t* __restrict__ pb, long long n, const float* __restrict__ ctrl,
                                                    float wd, float b1, float b2, float eps, float grad_scale) {
  if (ctrl[1] == 0.f) return;  // non-finite gradients: skip the step on every rank alike
  const float gs = ctrl[2] * grad_scale, lr = ctrl[3], bc1 = ctrl[4], bc2s = sqrtf(ctrl[5]);
  const long long stride = (long long)gridDim.x * blockDim.x * 4;
  for (long long e = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; e < n; e += stride) {
    // flat segments are padded to multiples of 64 elements, so e + 4 <= n always holds
    float4 pv = *reinterpret_cast<const float4*>(p + e);
    const float4 gv = *reinterpret_cast<const float4*>(g + e);
    float4 mv = *reinterpret_cast<const float4*>(m + e);
    float4 vv = *reinterpret_cast<const float4*>(v + e);
    float* pp = &pv.x;
    const float* gp = &gv.x;
    float* mp = &mv.x;
    float* vp = &vv.x;
    bf16x4 ob;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gg = gp[k] * gs;
      float w = pp[k] * (1.f - lr * wd);
      mp[k] = b1 * mp[k] + (1.f - b1) * gg;
      vp[k] = b2 * vp[k] + (1.f - b2) * gg * gg;
      const float denom = sqrtf(vp[k]) / bc2s + eps;
      w -= (lr / bc1) * (mp[k] / denom);
      pp[k] = w;
      ob[k] = (bf16_t)w;
    }
    *reinterpret_cast<float4*>(p + e) = pv;
    *reinterpret_cast<float4*>(m + e) = mv;
    *reinterpret_cast<float4*>(v + e) = vv;
    if (pb) *reinterpret_cast<bf16x4*>(pb + e) = ob;
  }
}

__global__ void cast_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, long long n) {
  const long long stride = (long long)gridDim.x * blockDim.x * 4;
  for (long long e = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; e < n; e += stride) {
    if (e + 4 <= n) {
      const float4 v = *reinterpret_cast<const float4*>(x + e);
      bf16x4 t;
      t[0] = (bf16_t)v.x; t[1] = (bf16_t)v.y; t[2] = (bf16_t)v.z; t[3] = (bf16_t)v.w;
      *reinterpret_cast<bf16x4*>(y + e) = t;
    } else {
      for (long long k = e; k < n; ++k) y[k] = (bf16_t)x[k];
    }
  }
}

// x [R, P] f32 -> y [rows_out, 3P] bf16, the hi/lo split of every element laid out along K so that ONE bf16 GEMM
// with K = 3P and f32 accumulation returns x . z to ~2^-17 relative (hi.hi + hi.lo + lo.hi; the lo.lo term is
// below f32 rounding of the sum): pattern 0 = [hi | hi | lo] (A side), pattern 1 = [hi | lo | hi] (B side).
// Rows R..rows_out-1 are zero (K padding of the gradient GEMMs).  One thread per 4 consecutive elements.
__global__ void split_bf16_kernel(const float* __restrict__ x, int R, int P, int rows_out, int pattern,
                                  bf16_t* __restrict__ y) {
  const long long total = (long long)rows_out * (P / 4);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(i / (P / 4)), c = (int)(i % (P / 4)) * 4;
    bf16x4 hi, lo;
    if (r < R) {
      const float4 v = *reinterpret_cast<const float4*>(x + (size_t)r * P + c);
      const float f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        hi[k] = (bf16_t)f[k];
        lo[k] = (bf16_t)(f[k] - (float)hi[k]);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) hi[k] = lo[k] = (bf16_t)0.f;
    }
    bf16_t* row = y + (size_t)r * 3 * P + c;
    *reinterpret_cast<bf16x4*>(row) = hi;
    *reinterpret_cast<bf16x4*>(row + P) = pattern ? lo : hi;
    *reinterpret_cast<bf16x4*>(row + 2 * P) = pattern ? hi : lo;
  }
}

__global__ void cast_f32_kernel(const bf16_t* __restrict__ x, float* __restrict__ y, long long n) {
  const long long stride = (long long)gridDim.x * blockDim.x * 8;
  for (long long e = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 8; e < n; e += stride) {
    if (e + 8 <= n) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + e);
      float4 a = make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
      float4 b = make_float4((float)v[4], (float)v[5], (float)v[6], (float)v[7]);
      *reinterpret_cast<float4*>(y + e) = a;
      *reinterpret_cast<float4*>(y + e + 4) = b;
    } else {
      for (long long k = e; k < n; ++k) y[k] = (float)x[k];
    }
  }
}

__global__ void axpy_kernel(const float* __restrict__ x, float alpha, float* __restrict__ y, long long n, int acc) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride)
    y[e] = acc ? y[e] + alpha * x[e] : alpha * x[e];
}

__global__ void gather_rows_bf16_kernel(const bf16_t* __restrict__ src, const int* __restrict__ map, int M, int H,
                                        bf16_t* __restrict__ dst) {
  const int chunks = H / 8;
  const long long total = (long long)M * chunks;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int m = (int)(i / chunks), c = (int)(i % chunks);
    *reinterpret_cast<u32x4*>(dst + (size_t)m * H + c * 8) =
        *reinterpret_cast<const u32x4*>(src + (size_t)map[m] * H + c * 8);
  }
}

// tok_lp[r] = logits[row_map[r], targets[r]] - logsumexp(logits[row_map[r], :V]); one wave per row.
// Compatibility path for callers that hold materialised [B,S,V] logits (reference model.py:1069-1079).
__global__ void logits_logprob_kernel(const float* __restrict__ logits, int ld, int V, const int* __restrict__ row_map,
                                      const long long* __restrict__ targets, int R, float* __restrict__ out) {
  const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= R) return;
  const float* row = logits + (size_t)(row_map ? row_map[r] : r) * ld;
  float mx = -INFINITY;
  for (int c = lane; c < V; c += 64) mx = fmaxf(mx, row[c]);
  mx = wave_max(mx);
  float s = 0.f;
  for (int c = lane; c < V; c += 64) s += __expf(row[c] - mx);
  s = wave_sum(s);
  if (lane == 0) out[r] = row[targets[r]] - (mx + __logf(s));
}

// Backward of the gather above on materialised logits: dlogits[row_map[r], :] = g[r] * (onehot(target) - softmax(row));
// one wave per scored row; rows that score no token keep the zeros the caller put there.
__global__ void logits_logprob_bwd_kernel(const float* __restrict__ logits, int ld, int V, const int* __restrict__ row_map,
                                          const long long* __restrict__ targets, const float* __restrict__ g, int R,
                                          float* __restrict__ dlogits) {
  const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= R) return;
  const size_t off = (size_t)(row_map ? row_map[r] : r) * ld;
  const float* row = logits + off;
  float* drow = dlogits + off;
  float mx = -INFINITY;
  for (int c = lane; c < V; c += 64) mx = fmaxf(mx, row[c]);
  mx = wave_max(mx);
  float s = 0.f;
  for (int c = lane; c < V; c += 64) s += __expf(row[c] - mx);
  s = wave_sum(s);
  const float lse = mx + __logf(s), gr = g[r];
  const int tgt = (int)targets[r];
  for (int c = lane; c < V; c += 64) drow[c] = gr * ((c == tgt ? 1.f : 0.f) - __expf(row[c] - lse));
}

inline int blocks_for(long long work, int per_block, int cap = 4096) {
  long long b = (work + per_block - 1) / per_block;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

#define REQUIRE(cond, who)                                 \
  if (!(cond)) {                                           \
    set_error(who ": bad arguments (" #cond ")");          \
    return PGCA_ERR_INVALID;                               \
  }

extern "C" int pgca_patchify(const float* pixels, int32_t B, int32_t image, int32_t patch, int32_t ld_out,
                             void* out_bf16, void* stream) {
  const int D = 3 * patch * patch;
  REQUIRE(pixels && out_bf16 && B > 0 && patch > 0 && image % patch == 0 && patch % 2 == 0 && image % 2 == 0 &&
              ld_out >= D && ld_out % 2 == 0 && (((uintptr_t)pixels & 15) == 0),
          "pgca_patchify");
  if (patch % 4 == 0 && image % 4 == 0 && ld_out == D) {
    const long long work = (long long)B * 3 * image * image / 4;
    hipLaunchKernelGGL(patchify_kernel, dim3(blocks_for(work, 256)), dim3(256), 0, (hipStream_t)stream, pixels, B,
                       image, patch, (bf16_t*)out_bf16);
  } else {
    const int G = image / patch;
    const long long work = (long long)B * G * G * (ld_out / 2);
    hipLaunchKernelGGL(patchify_pad_kernel, dim3(blocks_for(work, 256)), dim3(256), 0, (hipStream_t)stream, pixels, B,
                       image, patch, ld_out, (bf16_t*)out_bf16);
  }
  return check_launch("pgca_patchify");
}

extern "C" int pgca_vit_assemble(const float* patch_embeds, const float* cls, const float* pos, int32_t B, int32_t T,
                                 int32_t H, float* x, void* stream) {
  REQUIRE(patch_embeds && cls && pos && x && B > 0 && T > 1 && H > 0, "pgca_vit_assemble");
  hipLaunchKernelGGL(vit_assemble_kernel, dim3(blocks_for((long long)B * T * H, 256)), dim3(256), 0,
                     (hipStream_t)stream, patch_embeds, cls, pos, B, T, H, x);
  return check_launch("pgca_vit_assemble");
}

extern "C" int pgca_vit_assemble_bwd(const float* dx, int32_t B, int32_t T, int32_t H, void* dpatch_bf16, float* dcls,
                                     float* dpos, void* stream) {
  REQUIRE(dx && dpatch_bf16 && dcls && dpos && B > 0 && T > 1 && H > 0, "pgca_vit_assemble_bwd");
  hipLaunchKernelGGL(vit_assemble_bwd_kernel, dim3((T * H + 255) / 256), dim3(256), 0, (hipStream_t)stream, dx, B, T, H,
                     (bf16_t*)dpatch_bf16, dcls, dpos);
  return check_launch("pgca_vit_assemble_bwd");
}

extern "C" int pgca_seq_reduce(const float* tok_lp, const int32_t* seq_of_row, int32_t nrows, int32_t nseq,
                               const int32_t* seq_count, int32_t mode, float* seq_lp, void* stream) {
  REQUIRE(tok_lp && seq_of_row && seq_lp && nrows >= 0 && nseq > 0 && (!mode || seq_count), "pgca_seq_reduce");
  hipLaunchKernelGGL(seq_reduce_kernel, dim3((nseq + 3) / 4), dim3(256), 0, (hipStream_t)stream, tok_lp, seq_of_row,
                     nrows, nseq, seq_count, mode, seq_lp);
  return check_launch("pgca_seq_reduce");
}

extern "C" int pgca_seq_batch_prepare(const int64_t* ids, const int64_t* mask, int32_t Bq, int32_t S, int32_t* counts,
                                      int32_t* mask32, int32_t* row_map, int64_t* targets, int32_t* seq_of_row,
                                      int32_t* n_rows, void* stream) {
  REQUIRE(ids && mask && counts && row_map && targets && seq_of_row && n_rows && Bq > 0 && S > 1, "pgca_seq_batch_prepare");
  hipLaunchKernelGGL(seq_counts_kernel, dim3((Bq + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                     (const long long*)mask, Bq, S, counts, mask32);
  hipLaunchKernelGGL(seq_compact_kernel, dim3((Bq + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const long long*)ids,
                     (const long long*)mask, counts, Bq, S, row_map, (long long*)targets, seq_of_row, n_rows);
  return check_launch("pgca_seq_batch_prepare");
}

extern "C" int pgca_seq_pack_prepare(int32_t* mask32, int32_t Bq, int32_t S, int32_t pad_to, int32_t* lens, int32_t* cu,
                                     int32_t* row_ids, int32_t* n_packed, const int32_t* counts, int32_t* row_map,
                                     void* stream) {
  REQUIRE(mask32 && lens && cu && row_ids && n_packed && Bq > 0 && S > 0 && pad_to > 0 && (!row_map || counts),
          "pgca_seq_pack_prepare");
  hipLaunchKernelGGL(seq_lens_kernel, dim3((Bq + 3) / 4), dim3(256), 0, (hipStream_t)stream, mask32, Bq, S, lens);
  hipLaunchKernelGGL(seq_pack_kernel, dim3((Bq + 1 + 3) / 4), dim3(256), 0, (hipStream_t)stream, lens, Bq, S, pad_to,
                     cu, row_ids, n_packed, mask32, counts, row_map);
  return check_launch("pgca_seq_pack_prepare");
}

extern "C" int pgca_row_scale(const float* dseq, const int32_t* seq_of_row, const int32_t* seq_count, int32_t nrows,
                              int32_t mode, float* row_scale, void* stream) {
  REQUIRE(dseq && seq_of_row && row_scale && nrows > 0 && (!(mode & 1) || seq_count), "pgca_row_scale");
  hipLaunchKernelGGL(row_scale_kernel, dim3((nrows + 255) / 256), dim3(256), 0, (hipStream_t)stream, dseq, seq_of_row,
                     seq_count, nrows, mode, row_scale);
  return check_launch("pgca_row_scale");
}

extern "C" int pgca_dpo_loss(const float* pol_w, const float* pol_l, const float* ref_w, const float* ref_l, int32_t B,
                             float beta, float label_smoothing, float* loss, float* dpol_w, float* dpol_l,
                             float* metrics, void* stream) {
  REQUIRE(pol_w && pol_l && loss && B > 0 && ((!ref_w) == (!ref_l)), "pgca_dpo_loss");
  hipLaunchKernelGGL(dpo_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, pol_w, pol_l, ref_w, ref_l, B, beta,
                     label_smoothing, loss, dpol_w, dpol_l, metrics);
  return check_launch("pgca_dpo_loss");
}

extern "C" int pgca_masked_mean_fwd(const float* feats, const int32_t* mask, int32_t B, int32_t S, int32_t H,
                                    float* pooled, const int32_t* cu_seqlens, void* stream) {
  REQUIRE(feats && mask && pooled && B > 0 && S > 0 && H > 0, "pgca_masked_mean_fwd");
  hipLaunchKernelGGL(masked_mean_fwd_kernel, dim3((H + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, feats, mask,
                     S, H, pooled, cu_seqlens);
  return check_launch("pgca_masked_mean_fwd");
}
extern "C" int pgca_masked_mean_bwd(const float* dpooled, const int32_t* mask, int32_t B, int32_t S, int32_t H,
                                    float* dfeats, const int32_t* cu_seqlens, void* stream) {
  REQUIRE(dpooled && mask && dfeats && B > 0 && S > 0 && H > 0, "pgca_masked_mean_bwd");
  hipLaunchKernelGGL(masked_mean_bwd_kernel, dim3((H + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, dpooled, mask,
                     S, H, dfeats, cu_seqlens);
  return check_launch("pgca_masked_mean_bwd");
}

extern "C" int pgca_l2norm_fwd(const float* x, int32_t B, int32_t P, float* y, float* norm, void* stream) {
  REQUIRE(x && y && B > 0 && P > 0, "pgca_l2norm_fwd");
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, B, P, y, norm);
  return check_launch("pgca_l2norm_fwd");
}
extern "C" int pgca_l2norm_bwd(const float* dy, const float* y, const float* norm, int32_t B, int32_t P, float* dx,
                               void* stream) {
  REQUIRE(dy && y && norm && dx && B > 0 && P > 0, "pgca_l2norm_bwd");
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, dy, y, norm, B, P, dx);
  return check_launch("pgca_l2norm_bwd");
}

extern "C" int pgca_ntxent_loss(const float* lse_r, const float* lse_c, const float* diag, int32_t n_local,
                                int32_t n_total, float* loss, void* stream) {
  REQUIRE(lse_r && lse_c && diag && loss && n_local > 0 && n_total >= n_local, "pgca_ntxent_loss");
  hipLaunchKernelGGL(ntxent_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, lse_r, lse_c, diag, n_local,
                     n_total, loss);
  return check_launch("pgca_ntxent_loss");
}

extern "C" int pgca_sqnorm_blocks(int64_t n) { return (int)((n + SQ_BLOCK_ELEMS - 1) / SQ_BLOCK_ELEMS); }

extern "C" int pgca_sqnorm(const float* g, int64_t n, float* part, void* stream) {
  REQUIRE(g && part && n > 0 && (((uintptr_t)g & 15) == 0), "pgca_sqnorm");
  hipLaunchKernelGGL(sqnorm_kernel, dim3(pgca_sqnorm_blocks(n)), dim3(256), 0, (hipStream_t)stream, g, (long long)n,
                     part);
  return check_launch("pgca_sqnorm");
}

extern "C" int pgca_step_control(const float* part, int32_t nparts, float max_norm, float base_lr, int32_t warmup,
                                 int32_t total_steps, int32_t sched_stride, float beta1, float beta2, float grad_scale,
                                 const float* gate, float* ctrl, void* stream) {
  REQUIRE(part && ctrl && nparts > 0, "pgca_step_control");
  hipLaunchKernelGGL(step_control_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, part, nparts, max_norm, base_lr,
                     warmup, total_steps, sched_stride, beta1, beta2, grad_scale, gate, ctrl);
  return check_launch("pgca_step_control");
}

extern "C" int pgca_clip_coef(const float* part, int32_t nparts, float max_norm, float* coef, void* stream) {
  REQUIRE(part && coef && nparts > 0 && max_norm > 0.f, "pgca_clip_coef");
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, part, nparts, max_norm, coef);
  return check_launch("pgca_clip_coef");
}

extern "C" int pgca_scale_dev(float* x, int64_t n, const float* coef, void* stream) {
  REQUIRE(x && coef && n > 0 && (n % 4) == 0 && (((uintptr_t)x & 15) == 0), "pgca_scale_dev");
  hipLaunchKernelGGL(scale_dev_kernel, dim3(blocks_for(n / 4, 256, 8192)), dim3(256), 0, (hipStream_t)stream, x,
                     (long long)n, coef);
  return check_launch("pgca_scale_dev");
}

extern "C" int pgca_adamw(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, const float* ctrl,
                          float weight_decay, float beta1, float beta2, float eps, float grad_scale, void* stream) {
  REQUIRE(p && g && m && v && ctrl && n > 0 && (n % 4) == 0, "pgca_adamw");
  hipLaunchKernelGGL(adamw_kernel, dim3(blocks_for(n / 4, 256, 8192)), dim3(256), 0, (hipStream_t)stream, p, g, m, v,
                     (bf16_t*)p_bf16, (long long)n, ctrl, weight_decay, beta1, beta2, eps, grad_scale);
  return check_launch("pgca_adamw");
}

extern "C" int pgca_cast_bf16(const float* x, void* y_bf16, int64_t n, void* stream) {
  REQUIRE(x && y_bf16 && n > 0, "pgca_cast_bf16");
  hipLaunchKernelGGL(cast_bf16_kernel, dim3(blocks_for((n + 3) / 4, 256, 8192)), dim3(256), 0, (hipStream_t)stream, x,
                     (bf16_t*)y_bf16, (long long)n);
  return check_launch("pgca_cast_bf16");
}

extern "C" int pgca_cast_f32(const void* x_bf16, float* y, int64_t n, void* stream) {
  REQUIRE(x_bf16 && y && n > 0 && (((uintptr_t)x_bf16 & 15) == 0) && (((uintptr_t)y & 15) == 0), "pgca_cast_f32");
  hipLaunchKernelGGL(cast_f32_kernel, dim3(blocks_for((n + 7) / 8, 256, 8192)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x_bf16, y, (long long)n);
  return check_launch("pgca_cast_f32");
}

extern "C" int pgca_split_bf16(const float* x, int32_t R, int32_t P, int32_t rows_out, int32_t pattern, void* y_bf16,
                               void* stream) {
  REQUIRE(x && y_bf16 && R > 0 && P > 0 && (P % 8) == 0 && rows_out >= R && (pattern == 0 || pattern == 1) &&
              (((uintptr_t)x & 15) == 0),
          "pgca_split_bf16");
  hipLaunchKernelGGL(split_bf16_kernel, dim3(blocks_for((long long)rows_out * (P / 4), 256)), dim3(256), 0,
                     (hipStream_t)stream, x, R, P, rows_out, pattern, (bf16_t*)y_bf16);
  return check_launch("pgca_split_bf16");
}

extern "C" int pgca_axpy(const float* x, float alpha, float* y, int64_t n, int32_t accumulate, void* stream) {
  REQUIRE(x && y && n > 0, "pgca_axpy");
  hipLaunchKernelGGL(axpy_kernel, dim3(blocks_for(n, 256, 8192)), dim3(256), 0, (hipStream_t)stream, x, alpha, y,
                     (long long)n, accumulate);
  return check_launch("pgca_axpy");
}

extern "C" int pgca_gather_rows_bf16(const void* src, const int32_t* row_map, int32_t M, int32_t H, void* dst,
                                     void* stream) {
  REQUIRE(src && row_map && dst && M > 0 && H > 0 && (H % 8) == 0, "pgca_gather_rows_bf16");
  hipLaunchKernelGGL(gather_rows_bf16_kernel, dim3(blocks_for((long long)M * H / 8, 256)), dim3(256), 0,
                     (hipStream_t)stream, (const bf16_t*)src, row_map, M, H, (bf16_t*)dst);
  return check_launch("pgca_gather_rows_bf16");
}

extern "C" int pgca_logits_logprob(const float* logits, int32_t ld, int32_t V, const int32_t* row_map,
                                   const int64_t* targets, int32_t R, float* out, void* stream) {
  REQUIRE(logits && targets && out && R > 0 && V > 0 && ld >= V, "pgca_logits_logprob");
  hipLaunchKernelGGL(logits_logprob_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, ld, V,
                     row_map, (const long long*)targets, R, out);
  return check_launch("pgca_logits_logprob");
}

extern "C" int pgca_logits_logprob_bwd(const float* logits, int32_t ld, int32_t V, const int32_t* row_map,
                                       const int64_t* targets, const float* g, int32_t R, float* dlogits, void* stream) {
  REQUIRE(logits && targets && g && dlogits && ld >= V && V > 0 && R >= 0, "pgca_logits_logprob_bwd");
  if (R == 0) return PGCA_OK;
  hipLaunchKernelGGL(logits_logprob_bwd_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, ld, V,
                     row_map, (const long long*)targets, g, R, dlogits);
  return check_launch("pgca_logits_logprob_bwd");
}
