// Micro-benchmark: what can ONE workgroup (8 waves, one per CU) store per clock, in the access shape of the GEMM epilogue?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/store_bw tools/micro/store_bw.hip && /tmp/store_bw
// Each workgroup writes `reps` 256 x 256 tiles (bf16: 8 rows x 128 B per wave instruction, or f32: 4 rows x 256 B... see `mode`)
// into a row-major matrix of `ld` elements per row; clocks by s_memtime around the store loop (+ vmcnt(0)).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SEG>  // bytes per contiguous row segment written by one instruction: 128, 256, 512 or 1024
__global__ __launch_bounds__(512) void store_kernel(unsigned char* out, size_t ld_bytes, int rows_per_tile, int reps,
                                                    unsigned long long* clk, int rmw) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int LPR = SEG / 16;        // lanes per row segment
  constexpr int RPI = 64 / LPR;        // rows per instruction
  const int tile = blockIdx.x;
  unsigned long long t0, t1;
  f32x4 v = {1.f * lane, 2.f, 3.f, 4.f};
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int r = 0; r < reps; ++r) {
    // tile = rows_per_tile rows x 512 B (bf16 256 cols) per 'column block'; wave w owns rows [w*rows/8, ...)
    unsigned char* base = out + ((size_t)(tile * reps + r) * rows_per_tile) * ld_bytes;
    const int rows_w = rows_per_tile / 8;
    for (int cb = 0; cb < 512 / SEG; ++cb)
      for (int r0 = 0; r0 < rows_w; r0 += RPI) {
        const int row = wave * rows_w + r0 + lane / LPR;
        f32x4* p = reinterpret_cast<f32x4*>(base + (size_t)row * ld_bytes + cb * SEG + (lane % LPR) * 16);
        if (rmw) { f32x4 o = __builtin_nontemporal_load(p); v += o; }
        *p = v;
      }
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int SEG>
void run(const char* name, int grid, int reps, int rmw, unsigned char* buf, size_t ld_bytes, unsigned long long* dclk) {
  const int rows = 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  store_kernel<SEG><<<grid, 512>>>(buf, ld_bytes, rows, reps, dclk, rmw);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  store_kernel<SEG><<<grid, 512>>>(buf, ld_bytes, rows, reps, dclk, rmw);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> c(grid);
  hipMemcpy(c.data(), dclk, grid * 8, hipMemcpyDeviceToHost);
  double avg = 0; for (auto x : c) avg += x; avg /= grid;
  const double bytes = (double)rows * 512 * reps * (rmw ? 2 : 1);
  printf("%-28s grid %4d seg %4d B rmw %d: %7.0f clk/tile  %6.2f B/clk/CU   aggregate %6.2f TB/s (%.1f us)\n", name, grid, SEG, rmw,
         avg / reps, bytes / avg, bytes * grid / (ms * 1e-3) / 1e12, ms * 1e3);
}

int main() {
  const size_t ld_bytes = 3072 * 2;  // bf16 row of the qkv output
  const int reps = 16;
  const size_t total = (size_t)256 * reps * 256 * ld_bytes;  // 6.4 GB?  -> cap
  unsigned char* buf; unsigned long long* dclk;
  if (hipMalloc(&buf, total) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc(&dclk, 256 * 8);
  hipMemset(buf, 0, total);
  for (int grid : {1, 8, 64, 256}) {
    for (int rmw : {0, 1}) {
      run<128>("8 rows x 128 B / instr", grid, reps, rmw, buf, ld_bytes, dclk);
      run<256>("4 rows x 256 B / instr", grid, reps, rmw, buf, ld_bytes, dclk);
      run<512>("2 rows x 512 B / instr", grid, reps, rmw, buf, ld_bytes, dclk);
    }
  }
  return 0;
}
