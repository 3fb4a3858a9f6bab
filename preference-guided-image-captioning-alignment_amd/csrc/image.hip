// Image input transform on the device ("on-GPU resize/normalise"): decoded RGB uint8 HWC images ->
// Resize((S, S), bilinear, Pillow's antialiased two-pass fixed-point resample) -> ToTensor -> Normalize -> f32 [B, 3, S, S],
// BIT-EXACT with the reference's host path (data/preprocessing.py:44-48: torchvision Resize on a PIL image is
// PIL.Image.resize(BILINEAR); Pillow src/libImaging/Resample.c ImagingResampleHorizontal_8bpc / Vertical_8bpc).
//
// HBM-bound byte work: no matrix core involved.  Both passes move whole contiguous chunks with 16-byte / 32-bit accesses and
// transpose through LDS (kernel comments below).  The 22-bit fixed-point coefficient tables are computed once per
// (in, out) size on the host (input.resample_tables) exactly as Pillow does.
#include "common.h"

using namespace pgca;

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;  // Resample.c
// Taps are 22-bit fixed point (|k| <= 2^22 for the bilinear filter) and pixels 8-bit, so every product is a 24 x 24-bit
// multiply: __mul24 (v_mad_i32_i24, full rate) instead of the quarter-rate 32-bit integer multiply.

__device__ __forceinline__ int clip8(int v) {
  v >>= PRECISION_BITS;  // arithmetic shift, as the lookup index of the C code
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

constexpr int HROWS = 8;  // input rows per workgroup in pass 1 (fewer when the rows are too wide for the LDS staging)

// Pass 1.  One workgroup per HROWS consecutive input rows (rows of the whole batch are contiguous in both images and tmp):
// in [B*H, W, 3] u8 -> tmp [B*H, S, 3] u8.  The chunk is staged in LDS with 16-byte loads, every wave convolves its rows
// (one output pixel, three channels per lane) into an LDS image of the output chunk, which leaves with 16-byte stores.
__global__ __launch_bounds__(256) void resample_h_kernel(const unsigned char* __restrict__ in, int nrows, int W, int S,
                                                         const int* __restrict__ bounds, const int* __restrict__ coef,
                                                         int ksize, unsigned char* __restrict__ tmp, int in_lds,
                                                         int hrows) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* orow = lds + in_lds;
  const int r0 = blockIdx.x * hrows;
  const int nr = min(hrows, nrows - r0);
  const unsigned char* src = in + (size_t)r0 * W * 3;
  const int nbytes = nr * W * 3;
  // the chunk start is only byte-aligned in general: 16-byte loads from the aligned address at or below it
  const int mis = (int)((size_t)src & 15);
  const uint4* src16 = reinterpret_cast<const uint4*>(src - mis);
  const int nvec = (mis + nbytes + 15) >> 4;
  for (int i = threadIdx.x; i < nvec; i += blockDim.x) {
    // the last vector may run past the chunk: it stays inside the allocation except for the last chunk of the batch,
    // which ends bytewise
    if (i < nvec - 1 || blockIdx.x != gridDim.x - 1) {
      reinterpret_cast<uint4*>(lds)[i] = src16[i];
    } else {
      for (int b = 0; b < 16; ++b) {
        const int o = i * 16 + b;
        lds[o] = (o >= mis && o < mis + nbytes) ? src[o - mis] : 0;
      }
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int rr = wave; rr < nr; rr += 4) {
    const unsigned char* px = lds + mis + (size_t)rr * W * 3;
    unsigned char* o = orow + (size_t)rr * S * 3;
    for (int xx = lane; xx < S; xx += 64) {
      const int xmin = bounds[2 * xx], xmax = bounds[2 * xx + 1];
      const int* k = coef + (size_t)xx * ksize;
      int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
      for (int x = 0; x < xmax; ++x) {
        const int kv = k[x];
        const unsigned char* p = px + (x + xmin) * 3;
        s0 += __mul24((int)p[0], kv);
        s1 += __mul24((int)p[1], kv);
        s2 += __mul24((int)p[2], kv);
      }
      o[xx * 3 + 0] = (unsigned char)clip8(s0);
      o[xx * 3 + 1] = (unsigned char)clip8(s1);
      o[xx * 3 + 2] = (unsigned char)clip8(s2);
    }
  }
  __syncthreads();
  unsigned char* dst = tmp + (size_t)r0 * S * 3;
  const int obytes = nr * S * 3;
  if ((((size_t)dst | (size_t)obytes) & 15) == 0) {
    for (int i = threadIdx.x; i < (obytes >> 4); i += blockDim.x)
      reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(orow)[i];
  } else {
    for (int i = threadIdx.x; i < obytes; i += blockDim.x) dst[i] = orow[i];
  }
}

// Pass 2.  One workgroup per output row (b, yy): tmp [B, H, S, 3] u8 -> out f32 [B, 3, S, S] (+ optional resized u8
// [B, S, S, 3]).  The vertical convolution is independent per BYTE of the interleaved row, so a thread takes four
// consecutive bytes with one 32-bit load per tap (S % 4 == 0; otherwise byte by byte); the u8 row is transposed through
// LDS so that the three f32 planes are written with consecutive lanes on consecutive addresses.
__global__ __launch_bounds__(256) void resample_v_norm_kernel(const unsigned char* __restrict__ tmp, int H, int S,
                                                              const int* __restrict__ bounds,
                                                              const int* __restrict__ coef, int ksize, float m0, float m1,
                                                              float m2, float d0, float d1, float d2,
                                                              unsigned char* __restrict__ resized,
                                                              float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char urow[];
  const int yy = blockIdx.x;
  const size_t b = blockIdx.y;
  const int ymin = bounds[2 * yy], ymax = bounds[2 * yy + 1];
  const int* k = coef + (size_t)yy * ksize;
  const int rowb = S * 3;
  const unsigned char* src = tmp + (b * H + ymin) * (size_t)rowb;
  if ((S & 3) == 0) {
    for (int j = threadIdx.x * 4; j < rowb; j += blockDim.x * 4) {
      int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0, a3 = a0;
      for (int y = 0; y < ymax; ++y) {
        const int kv = k[y];
        const unsigned v = *reinterpret_cast<const unsigned*>(src + (size_t)y * rowb + j);
        a0 += __mul24((int)(v & 255u), kv);
        a1 += __mul24((int)((v >> 8) & 255u), kv);
        a2 += __mul24((int)((v >> 16) & 255u), kv);
        a3 += __mul24((int)(v >> 24), kv);
      }
      *reinterpret_cast<unsigned*>(urow + j) =
          (unsigned)clip8(a0) | ((unsigned)clip8(a1) << 8) | ((unsigned)clip8(a2) << 16) | ((unsigned)clip8(a3) << 24);
    }
  } else {
    for (int j = threadIdx.x; j < rowb; j += blockDim.x) {
      int a0 = 1 << (PRECISION_BITS - 1);
      for (int y = 0; y < ymax; ++y) a0 += __mul24((int)src[(size_t)y * rowb + j], k[y]);
      urow[j] = (unsigned char)clip8(a0);
    }
  }
  __syncthreads();
  if (resized) {
    unsigned char* q = resized + (b * S + yy) * (size_t)rowb;
    for (int j = threadIdx.x; j < rowb; j += blockDim.x) q[j] = urow[j];
  }
  // ToTensor: float32(v) / 255; Normalize: (t - mean) / std - correctly rounded IEEE divisions, as torch on the host
  const size_t plane = (size_t)S * S;
  float* o = out + b * 3 * plane + (size_t)yy * S;
  for (int xx = threadIdx.x; xx < S; xx += blockDim.x) {
    const unsigned char* p = urow + xx * 3;
    o[xx] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)p[0], 255.0f), m0), d0);
    o[plane + xx] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)p[1], 255.0f), m1), d1);
    o[2 * plane + xx] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)p[2], 255.0f), m2), d2);
  }
}

}  // namespace

extern "C" int pgca_image_preprocess(const uint8_t* images, int32_t B, int32_t H, int32_t W, int32_t S,
                                     const int32_t* xbounds, const int32_t* xcoef, int32_t xk, const int32_t* ybounds,
                                     const int32_t* ycoef, int32_t yk, float mean0, float mean1, float mean2, float std0,
                                     float std1, float std2, uint8_t* tmp, uint8_t* resized_u8, float* out, void* stream) {
  if (!images || !xbounds || !xcoef || !ybounds || !ycoef || !tmp || !out || B <= 0 || H <= 0 || W <= 0 || S <= 0 ||
      xk <= 0 || yk <= 0 || B > 65535 || (long long)B * H > 0x7fffffffLL) {
    set_error("pgca_image_preprocess: bad arguments");
    return PGCA_ERR_INVALID;
  }
  const int nrows = B * H;
  int hrows = HROWS;
  size_t in_lds = 0, out_lds = 0;
  for (;; hrows >>= 1) {
    in_lds = ((size_t)hrows * W * 3 + 15 + 48) & ~(size_t)15;   // slack: alignment shift + the word pair of the last tap
    out_lds = ((size_t)hrows * S * 3 + 15) & ~(size_t)15;
    if (in_lds + out_lds <= 64 * 1024 || hrows == 1) break;
  }
  const size_t v_lds = ((size_t)S * 3 + 15) & ~(size_t)15;
  if (in_lds + out_lds > 64 * 1024 || v_lds > 64 * 1024) {
    set_error("pgca_image_preprocess: a row of %d -> %d pixels does not fit the 64 KiB LDS staging", W, S);
    return PGCA_ERR_INVALID;
  }
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(resample_h_kernel, dim3((nrows + hrows - 1) / hrows), dim3(256), in_lds + out_lds, s, images, nrows, W,
                     S, xbounds, xcoef, xk, tmp, (int)in_lds, hrows);
  hipLaunchKernelGGL(resample_v_norm_kernel, dim3(S, B), dim3(256), v_lds, s, tmp, H, S, ybounds, ycoef, yk, mean0, mean1,
                     mean2, std0, std1, std2, resized_u8, out);
  return check_launch("pgca_image_preprocess");
}
