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


// ------------------------------------------------------------------------------------------------------------------
// Training transform on the device (reference data/preprocessing.py:52-70, augment=True):
//   RandomResizedCrop -> RandomHorizontalFlip -> ColorJitter -> RandomRotation(5) -> ToTensor -> Normalize,
// BIT-EXACT with torchvision's PIL backend given the same random draws (input.draw_train_params makes them on the host;
// oracle/image_restatement.py restates each op and is pinned against Pillow itself).  Three launches per batch:
//   1/2. crop + Pillow's two-pass resample with PER-IMAGE boxes and tap tables (the crop size differs per image),
//   3.   one workgroup per image: the S x S x 3 uint8 image lives in LDS through flip, the four jitter operations in their
//        drawn order (Blend.c float32 blends against black / the mean grey / the per-pixel grey; the hue turn through
//        Convert.c's HSV round trip) and the nearest-neighbour rotation gather (Geometry.c affine_fixed, 16.16 fixed
//        point), then ToTensor + Normalize straight into the f32 planes.
// Byte and integer work plus a little float arithmetic that must round as the C library's does: contraction is off.
constexpr int TP_INTS = 20;  // per image: i, j, h, w, flip, order[4], hue shift, rotate?, a0..a5, pad[3]

__global__ __launch_bounds__(256) void crop_resample_h_kernel(const unsigned char* __restrict__ in, int H, int W, int S,
                                                              const int* __restrict__ params,
                                                              const int* __restrict__ bounds,
                                                              const int* __restrict__ coef, int ksize,
                                                              unsigned char* __restrict__ tmp) {
  const int b = blockIdx.y, y = blockIdx.x;
  const int* tp = params + (size_t)b * TP_INTS;
  const int ci = tp[0], cj = tp[1], ch = tp[2];
  if (y >= ch) return;
  const unsigned char* row = in + (((size_t)b * H + ci + y) * W + cj) * 3;
  unsigned char* o = tmp + ((size_t)b * H + y) * S * 3;
  for (int xx = threadIdx.x; xx < S; xx += blockDim.x) {
    const int xmin = bounds[((size_t)b * S + xx) * 2], xmax = bounds[((size_t)b * S + xx) * 2 + 1];
    const int* k = coef + ((size_t)b * S + xx) * ksize;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int x = 0; x < xmax; ++x) {
      const int kv = k[x];
      const unsigned char* p = row + (x + xmin) * 3;
      s0 += __mul24((int)p[0], kv);
      s1 += __mul24((int)p[1], kv);
      s2 += __mul24((int)p[2], kv);
    }
    o[xx * 3 + 0] = (unsigned char)clip8(s0);
    o[xx * 3 + 1] = (unsigned char)clip8(s1);
    o[xx * 3 + 2] = (unsigned char)clip8(s2);
  }
}

__global__ __launch_bounds__(256) void crop_resample_v_kernel(const unsigned char* __restrict__ tmp, int H, int S,
                                                              const int* __restrict__ bounds,
                                                              const int* __restrict__ coef, int ksize,
                                                              unsigned char* __restrict__ resized) {
  const int yy = blockIdx.x;
  const size_t b = blockIdx.y;
  const int ymin = bounds[(b * S + yy) * 2], ymax = bounds[(b * S + yy) * 2 + 1];
  const int* k = coef + (b * S + yy) * ksize;
  const int rowb = S * 3;
  const unsigned char* src = tmp + (b * H + ymin) * (size_t)rowb;
  unsigned char* o = resized + (b * S + yy) * (size_t)rowb;
  for (int j = threadIdx.x; j < rowb; j += blockDim.x) {
    int a0 = 1 << (PRECISION_BITS - 1);
    for (int y = 0; y < ymax; ++y) a0 += __mul24((int)src[(size_t)y * rowb + j], k[y]);
    o[j] = (unsigned char)clip8(a0);
  }
}

// Float arithmetic below must round as the host C library's does, one operation at a time.  hipcc contracts a * b + c
// into one fma by default and HIP's __fmul_rn / __fadd_rn / __dmul_rn ... are plain operators compiled under that default
// (their bodies fuse again after inlining), so these functions use the bare operators under `fp contract(off)`.

// Blend.c: (UINT8)(in1 + alpha * (in2 - in1)) in float32, two roundings; clipped only when extrapolating
__device__ __forceinline__ int blend8(int in1, int in2, float alpha, bool interp) {
#pragma clang fp contract(off)
  const float prod = alpha * (float)(in2 - in1);
  const float t = (float)in1 + prod;
  if (interp) return (int)t;
  return t <= 0.f ? 0 : (t >= 255.f ? 255 : (int)t);
}
__device__ __forceinline__ int grey8(int r, int g, int b) {  // Convert.c L24
  return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16;
}
__device__ __forceinline__ int clamp255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// Convert.c rgb2hsv -> uint8 wrap-around hue turn -> hsv2rgb
__device__ void hue_turn(int& r, int& g, int& b, int shift) {
#pragma clang fp contract(off)
  const int maxc = max(r, max(g, b)), minc = min(r, min(g, b));
  int uh = 0, us = 0;
  const int uv = maxc;
  if (minc != maxc) {
    const float cr = (float)(maxc - minc);
    const float s = cr / (float)maxc;
    const float rc = (float)(maxc - r) / cr, gc = (float)(maxc - g) / cr, bc = (float)(maxc - b) / cr;
    float h;
    if (r == maxc) {
      h = bc - gc;
    } else if (g == maxc) {
      const double t2 = 2.0 + (double)rc;
      h = (float)(t2 - (double)bc);
    } else {
      const double t4 = 4.0 + (double)gc;
      h = (float)(t4 - (double)rc);
    }
    const double h6 = (double)h / 6.0;
    double hd = h6 + 1.0;              // in (0, 2): fmod(., 1.0) is an exact floor removal
    hd = hd - floor(hd);
    h = (float)hd;
    const double h255 = (double)h * 255.0, s255 = (double)s * 255.0;
    uh = clamp255((int)h255);
    us = clamp255((int)s255);
  }
  uh = (uh + shift) & 255;
  if (us == 0) {
    r = g = b = uv;
    return;
  }
  const double x6 = (double)uh * 6.0;
  const double x = x6 / 255.0;
  const double fi = floor(x);
  const double f = x - fi;
  const double fs = (double)us / 255.0;
  const double vf = (double)uv;
  const double one_f = 1.0 - f;
  const double fsf = fs * f, fsg = fs * one_f;
  const double mp = 1.0 - fs, mq = 1.0 - fsf, mt = 1.0 - fsg;
  const double vp = vf * mp, vq = vf * mq, vt = vf * mt;
  const double rp = vp + 0.5, rq = vq + 0.5, rt = vt + 0.5;   // C round(): half away from zero, values >= 0
  const int p = clamp255((int)floor(rp)), q = clamp255((int)floor(rq)), t = clamp255((int)floor(rt));
  switch (((int)fi) % 6) {
    case 0: r = uv; g = t; b = p; break;
    case 1: r = q; g = uv; b = p; break;
    case 2: r = p; g = uv; b = t; break;
    case 3: r = p; g = q; b = uv; break;
    case 4: r = t; g = p; b = uv; break;
    default: r = uv; g = p; b = q; break;
  }
}

constexpr int AUG_THREADS = 1024;

__global__ __launch_bounds__(AUG_THREADS) void augment_kernel(const unsigned char* __restrict__ resized, int S,
                                                              const int* __restrict__ params,
                                                              const float* __restrict__ factors, float m0, float m1,
                                                              float m2, float d0, float d1, float d2,
                                                              unsigned char* __restrict__ aug_u8,
                                                              float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char img[];  // [S * S * 3] then the reduction words
  const int npx = S * S;
  unsigned* red = reinterpret_cast<unsigned*>(img + (((size_t)npx * 3 + 15) & ~(size_t)15));  // [16 waves + 1]
  const size_t b = blockIdx.x;
  const int* tp = params + b * TP_INTS;
  const float* fc = factors + b * 3;
  const int t = threadIdx.x;
  const unsigned char* src = resized + b * (size_t)npx * 3;
  const int flip = tp[4];
  // pixels t, t + 1024, ... belong to this thread through every pixel-local operation (no barrier between those)
  for (int i = t; i < npx; i += AUG_THREADS) {
    const int y = i / S, x = i - y * S;
    const unsigned char* p = src + ((size_t)y * S + (flip ? S - 1 - x : x)) * 3;
    img[i * 3 + 0] = p[0];
    img[i * 3 + 1] = p[1];
    img[i * 3 + 2] = p[2];
  }
  for (int step = 0; step < 4; ++step) {
    const int op = tp[5 + step];
    if (op == 0) {         // ImageEnhance.Brightness: blend(black, image, f)
      const float f = fc[0];
      const bool interp = f >= 0.f && f <= 1.f;
      for (int i = t; i < npx; i += AUG_THREADS)
        for (int c = 0; c < 3; ++c) img[i * 3 + c] = (unsigned char)blend8(0, img[i * 3 + c], f, interp);
    } else if (op == 1) {  // ImageEnhance.Contrast: blend(mean grey, image, f), mean = int(sum(L) / n + 0.5)
      unsigned part = 0;
      for (int i = t; i < npx; i += AUG_THREADS) part += (unsigned)grey8(img[i * 3], img[i * 3 + 1], img[i * 3 + 2]);
      for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
      __syncthreads();                       // the previous use of red[] is over
      if ((t & 63) == 0) red[t >> 6] = part;
      __syncthreads();
      if (t == 0) {
        unsigned long long sum = 0;
        for (int w = 0; w < AUG_THREADS / 64; ++w) sum += red[w];
        red[16] = (unsigned)(int)((double)sum / (double)npx + 0.5);   // (a division and an add: nothing to fuse)
      }
      __syncthreads();
      const int mean = (int)red[16];
      const float f = fc[1];
      const bool interp = f >= 0.f && f <= 1.f;
      for (int i = t; i < npx; i += AUG_THREADS)
        for (int c = 0; c < 3; ++c) img[i * 3 + c] = (unsigned char)blend8(mean, img[i * 3 + c], f, interp);
    } else if (op == 2) {  // ImageEnhance.Color: blend(grey(image), image, f)
      const float f = fc[2];
      const bool interp = f >= 0.f && f <= 1.f;
      for (int i = t; i < npx; i += AUG_THREADS) {
        const int r = img[i * 3], g = img[i * 3 + 1], bb = img[i * 3 + 2];
        const int l = grey8(r, g, bb);
        img[i * 3 + 0] = (unsigned char)blend8(l, r, f, interp);
        img[i * 3 + 1] = (unsigned char)blend8(l, g, f, interp);
        img[i * 3 + 2] = (unsigned char)blend8(l, bb, f, interp);
      }
    } else {               // adjust_hue
      const int shift = tp[9];
      for (int i = t; i < npx; i += AUG_THREADS) {
        int r = img[i * 3], g = img[i * 3 + 1], bb = img[i * 3 + 2];
        hue_turn(r, g, bb, shift);
        img[i * 3 + 0] = (unsigned char)r;
        img[i * 3 + 1] = (unsigned char)g;
        img[i * 3 + 2] = (unsigned char)bb;
      }
    }
  }
  __syncthreads();  // the rotation gathers other threads' pixels
  const int rot = tp[10], a0 = tp[11], a1 = tp[12], a2 = tp[13], a3 = tp[14], a4 = tp[15], a5 = tp[16];
  float* o = out + b * 3 * (size_t)npx;
  for (int i = t; i < npx; i += AUG_THREADS) {
    int r, g, bb;
    int si = i;
    if (rot) {
      const int y = i / S, x = i - y * S;
      const int xin = (a2 + a1 * y + a0 * x) >> 16, yin = (a5 + a4 * y + a3 * x) >> 16;
      si = (xin >= 0 && xin < S && yin >= 0 && yin < S) ? yin * S + xin : -1;
    }
    if (si >= 0) {
      r = img[si * 3];
      g = img[si * 3 + 1];
      bb = img[si * 3 + 2];
    } else {
      r = g = bb = 0;  // fillcolor
    }
    if (aug_u8) {
      unsigned char* q = aug_u8 + (b * npx + i) * 3;
      q[0] = (unsigned char)r;
      q[1] = (unsigned char)g;
      q[2] = (unsigned char)bb;
    }
    o[i] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)r, 255.0f), m0), d0);
    o[npx + i] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)g, 255.0f), m1), d1);
    o[2 * (size_t)npx + i] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)bb, 255.0f), m2), d2);
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

extern "C" int pgca_image_train_transform(const uint8_t* images, int32_t B, int32_t H, int32_t W, int32_t S,
                                          const int32_t* params, const float* factors, const int32_t* xbounds,
                                          const int32_t* xcoef, int32_t xk, const int32_t* ybounds, const int32_t* ycoef,
                                          int32_t yk, float mean0, float mean1, float mean2, float std0, float std1,
                                          float std2, uint8_t* tmp, uint8_t* resized_u8, uint8_t* aug_u8, float* out,
                                          void* stream) {
  if (!images || !params || !factors || !xbounds || !xcoef || !ybounds || !ycoef || !tmp || !resized_u8 || !out ||
      B <= 0 || H <= 0 || W <= 0 || S <= 0 || xk <= 0 || yk <= 0 || B > 65535 || H > 65535 ||
      (long long)B * H > 0x7fffffffLL) {
    set_error("pgca_image_train_transform: bad arguments");
    return PGCA_ERR_INVALID;
  }
  const size_t lds = (((size_t)S * S * 3 + 15) & ~(size_t)15) + 17 * sizeof(unsigned);
  if (lds > 160 * 1024) {
    set_error("pgca_image_train_transform: a %d x %d image does not fit the 160 KiB LDS (image_size <= 230)", S, S);
    return PGCA_ERR_INVALID;
  }
  static const hipError_t attr =
      hipFuncSetAttribute((const void*)augment_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (attr != hipSuccess) {
    set_error("pgca_image_train_transform: cannot raise dynamic LDS limit");
    return PGCA_ERR_LAUNCH;
  }
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(crop_resample_h_kernel, dim3(H, B), dim3(256), 0, s, images, H, W, S, params, xbounds, xcoef, xk,
                     tmp);
  hipLaunchKernelGGL(crop_resample_v_kernel, dim3(S, B), dim3(256), 0, s, tmp, H, S, ybounds, ycoef, yk, resized_u8);
  hipLaunchKernelGGL(augment_kernel, dim3(B), dim3(AUG_THREADS), lds, s, resized_u8, S, params, factors, mean0, mean1,
                     mean2, std0, std1, std2, aug_u8, out);
  return check_launch("pgca_image_train_transform");
}
