// Phase-staggered 256 x 256 GEMM main loop for gfx950 (the "8-phase" idea of the CDNA4 programming guide, restated
// for this library's LDS images).  Eight waves; the two waves of every SIMD belong to different GROUPS (rows 0-127 /
// 128-255 of the tile) that run the same schedule ONE BARRIER APART:
//     group 0:  L0 | M0 | L1 | M1 | L0 | M0 ...      L = fragment reads + 2 LDS-DMA pieces + waits  (no MFMA)
//     group 1:       L0 | M0 | L1 | M1 | L0 ...      M = 16 MFMAs at raised priority                ( | = s_barrier )
// so while one wave of a SIMD is stuck issuing its copy instructions (~40-60 clocks each) or waiting for LDS, the
// other owns the matrix pipe.  K advances in 32-deep tiles through FOUR LDS stages; the copy of tile t+3 is issued
// during tile t (A pieces in L1, B pieces in the next L0) and retired with a COUNTED s_waitcnt vmcnt two barriers
// before its first read - never vmcnt(0) in the loop.
//   visibility: a wave's pieces of tile t+1 are retired in t.L1; t+1.L0 reads them after two more barriers (one for
//               the data, one because the other group runs a barrier late);
//   WAR:        a stage is re-filled >= 2 barriers after the last read of it was retired (lgkmcnt(0) opens every M).
#include "gemm_device.h"

namespace {

constexpr int PBK = 32, PSTAGES = 4;
constexpr int PTILE_BYTES = 256 * PBK * 2;  // 16 KiB per operand per stage

__device__ __forceinline__ int swz4p(int q) { return (0x78 >> (2 * q)) & 3; }  // {0,2,3,1}

template <int KS>
struct DmaP {
  unsigned goff[2];
  __device__ __forceinline__ void init(int lane, int wave, int ld, int origin, int extent) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int j = wave * 2 + i;  // 1-KiB piece of the 16-KiB tile image
      if (KS == 0) {               // [256 rows][32 k]: piece = 16 rows x 64 B; chunk c of row r at c ^ swz4p((r>>2)&3)
        const int r = 16 * j + (lane >> 2);
        const int c = (lane & 3) ^ swz4p((lane >> 4) & 3);
        const int rg = min(origin + r, extent - 1) - origin;
        goff[i] = (unsigned)(rg * ld + c * 8) * 2u;
      } else {                     // [32 k][256 cols]: piece = 2 k-rows x 512 B (same image as the BK = 64 kernel)
        const int k = 2 * j + (lane >> 5);
        const int c16 = lane & 31;
        const int h = (k & 3) | (((k >> 3) & 1) << 2);
        const int col = (((c16 >> 1) ^ h) << 4) + ((c16 & 1) << 3);
        const int cg = min(origin + col, extent - 8) - origin;
        goff[i] = (unsigned)(k * ld + cg) * 2u;
      }
    }
  }
  __device__ __forceinline__ void issue(const bf16_t* base, unsigned char* tile, int wave) const {
    const unsigned long long b = (unsigned long long)base;
    u32x4 rs;
    rs[0] = (unsigned)b;
    rs[1] = (unsigned)(b >> 32) & 0xffffu;
    rs[2] = 0x7ffffff0u;
    rs[3] = 0x00020000u;
    const unsigned lds0 = (unsigned)(size_t)LDS_PTR(tile) + (unsigned)wave * 2048u;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                   :
                   : "s"(lds0 + i * 1024u), "v"(goff[i]), "s"(rs)
                   : "memory");
    }
  }
};

template <int KS>
__device__ __forceinline__ bf16x8 read_frag_p(const unsigned char* lds, int wbase, int sub, int lane) {
  if (KS == 0) {
    const int row = wbase + sub * 16 + (lane & 15);
    const int pos = (lane >> 4) ^ swz4p((lane >> 2) & 3);
    return *reinterpret_cast<const bf16x8*>(lds + row * 64 + pos * 16);
  } else {
    return read_frag<1, 512>(lds, wbase, sub, 0, lane);
  }
}


template <int N>
__device__ __forceinline__ void wait_vm_p() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// -DPGCA_GEMM_TIMING: main-loop / epilogue clocks per workgroup; -DPGCA_GEMM_TIMING2 adds per-phase clocks
// (tools/gemm_bench.py --timing reads them from the stat_max / stat_sum buffers).  Stamps cost a lgkmcnt(0) each.
#ifdef PGCA_GEMM_TIMING2
#define PSTAMP(v) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory")
#else
#define PSTAMP(v)
#endif
// Diagnostic fillers (-DPGCA_FILL_L=n / -DPGCA_FILL_M=n): n independent VALU instructions per L phase (after its copy
// instructions, before the barrier) / per MFMA gap.  They calibrate how much epilogue work of a PREVIOUS tile the main
// loop could carry for free (tools/gemm_fill_probe.sh); never defined in the product build.
#ifndef PGCA_FILL_L
#define PGCA_FILL_L 0
#endif
#ifndef PGCA_FILL_M
#define PGCA_FILL_M 0
#endif
template <int N>
__device__ __forceinline__ void valu_fill(float (&d)[8]) {
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(d[i & 7]) : "v"(d[(i + 3) & 7]));
}
#define PGCA_PBAR()                       \
  do {                                    \
    __builtin_amdgcn_sched_barrier(0);    \
    __builtin_amdgcn_s_barrier();         \
    __builtin_amdgcn_sched_barrier(0);    \
  } while (0)

template <int LA, int LB>
__global__ __launch_bounds__(512, 2) void gemm256s_kernel(const pgca_gemm_args a, int ntm, int ntn, int nk_per_split,
                                                          int stagger) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smems[];  // [4 stages][A 16 KiB | B 16 KiB]
#ifdef PGCA_GEMM_TIMING
  unsigned long long ts0, tr0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts0)::"memory");
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tr0)::"memory");   // constant 100 MHz
#endif

  // De-phasing: with 128 KiB of LDS one workgroup owns a CU, every CU starts its tile at the same time and all 256
  // epilogues hit HBM together (store bursts at the HBM rate while the matrix pipes idle, then the reverse).  The
  // FIRST wave of workgroups (one per CU) therefore starts in four time slots `stagger` x 1024 clocks apart; later
  // workgroups inherit their CU's offset, so epilogue traffic of one quarter of the chip overlaps the main loops of
  // the rest.  Costs 3 x stagger x 1024 clocks once per launch (the tail), so it is used on many-round launches only.
  if (stagger > 0 && gridDim.y == 1 && blockIdx.x < 256) {
    const int slot = (blockIdx.x >> 3) & 3;
    for (int i = 0; i < slot * stagger; ++i) __builtin_amdgcn_s_sleep(16);  // 16 x 64 clocks
  }

  const int nwg = ntm * ntn;
  int bid = blockIdx.x;
  {
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  constexpr int GROUP_M = 8;
  const int per_group = GROUP_M * ntn;
  const int group = bid / per_group;
  const int first_m = group * GROUP_M;
  const int gsize = min(GROUP_M, ntm - first_m);
  const int in_group = bid - group * per_group;
  const int tm = first_m + in_group % gsize, tn = in_group / gsize;
  const int m0 = tm * BM2, n0 = tn * BN2;

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 2, wn = wave & 3;  // wm = group: waves w and w + 4 share a SIMD

  const bf16_t* A = reinterpret_cast<const bf16_t*>(a.A);
  const bf16_t* B = reinterpret_cast<const bf16_t*>(a.B);

  DmaP<LA> da;
  DmaP<LB> db;
  da.init(lane, wave, a.lda, m0, LA == 0 ? a.M : ((a.M + 7) & ~7));
  db.init(lane, wave, a.ldb, n0, LB == 0 ? a.N : ((a.N + 7) & ~7));
  const bf16_t* abase = LA == 0 ? A + (size_t)m0 * a.lda : A + m0;
  const bf16_t* bbase = LB == 0 ? B + (size_t)n0 * a.ldb : B + n0;
  const size_t astep = LA == 0 ? (size_t)PBK : (size_t)PBK * a.lda;
  const size_t bstep = LB == 0 ? (size_t)PBK : (size_t)PBK * a.ldb;

  f32x4 acc[2][4][4];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[h][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int kt0 = blockIdx.y * nk_per_split * 2;
  const int nk = min(nk_per_split * 2, a.K / PBK - kt0);  // even, >= 2
  abase += (size_t)kt0 * astep;
  bbase += (size_t)kt0 * bstep;

  // prologue: tiles 0 and 1 whole, A of tile 2 (its B goes out in the first L0): 2 instructions per operand piece set
  da.issue(abase, smems, wave);
  db.issue(bbase, smems + PTILE_BYTES, wave);
  da.issue(abase + astep, smems + 2 * PTILE_BYTES, wave);
  db.issue(bbase + bstep, smems + 3 * PTILE_BYTES, wave);
  if (2 < nk) {
    da.issue(abase + 2 * astep, smems + 4 * PTILE_BYTES, wave);
    wait_vm_p<6>();  // tile 0 landed
  } else {
    wait_vm_p<4>();
  }
  PGCA_PBAR();
  PGCA_PBAR();                 // second barrier: same distance (two) as in the steady state
  if (wm == 1) PGCA_PBAR();    // group 1 runs one barrier late from here on

#ifdef PGCA_GEMM_TIMING
  unsigned long long ts1, ts2, ts3;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts1)::"memory");
#endif
#ifdef PGCA_GEMM_TIMING2
  unsigned long long q0 = 0, q1 = 0, q2 = 0, q3 = 0, q4 = 0, q5 = 0, q6 = 0, q7 = 0, q8 = 0;
  unsigned long long aL0 = 0, aB1 = 0, aM0 = 0, aB2 = 0, aL1 = 0, aB3 = 0, aM1 = 0, aB4 = 0;
#endif
  float fill[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
  for (int kt = 0; kt < nk; ++kt) {
    PSTAMP(q0);
    const unsigned char* la = smems + (kt & 3) * 2 * PTILE_BYTES;
    const unsigned char* lb = la + PTILE_BYTES;
    bf16x8 fa[4], fb[4];
    // ---------------- L0: B fragments + A rows 0-63 of this wave's 128; B pieces of tile kt+2
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[j] = read_frag_p<LB>(lb, wn * 64, j, lane);
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i] = read_frag_p<LA>(la, wm * 128, i, lane);
    __builtin_amdgcn_sched_barrier(0);
    if (kt >= 1 && kt + 2 < nk) db.issue(bbase + (size_t)(kt + 2) * bstep, smems + ((kt + 2) & 3) * 2 * PTILE_BYTES + PTILE_BYTES, wave);
    if (kt == 0 && 2 < nk) db.issue(bbase + 2 * bstep, smems + 5 * PTILE_BYTES, wave);
    if (PGCA_FILL_L) valu_fill<PGCA_FILL_L>(fill);
    PSTAMP(q1);
    PGCA_PBAR();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // after the barrier: the fragment latency hides in the barrier wait
    __builtin_amdgcn_sched_barrier(0);
    PSTAMP(q2);
    // ---------------- M0
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[0][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[0][i][j], 0, 0, 0);
        if (PGCA_FILL_M) valu_fill<PGCA_FILL_M>(fill);
      }
    __builtin_amdgcn_s_setprio(0);
    PSTAMP(q3);
    PGCA_PBAR();
    PSTAMP(q4);
    // ---------------- L1: A rows 64-127; A pieces of tile kt+3; retire tile kt+1
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i] = read_frag_p<LA>(la, wm * 128 + 64, i, lane);
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 3 < nk) {
      da.issue(abase + (size_t)(kt + 3) * astep, smems + ((kt + 3) & 3) * 2 * PTILE_BYTES, wave);
      wait_vm_p<6>();  // in flight: A(kt+2) B(kt+2) A(kt+3); everything up to B(kt+1) has landed
    } else if (kt + 2 < nk) {
      wait_vm_p<4>();  // A(kt+2) B(kt+2)
    } else {
      wait_vm_p<0>();
    }
    if (PGCA_FILL_L) valu_fill<PGCA_FILL_L>(fill);
    PSTAMP(q5);
    PGCA_PBAR();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    PSTAMP(q6);
    // ---------------- M1
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[1][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[1][i][j], 0, 0, 0);
        if (PGCA_FILL_M) valu_fill<PGCA_FILL_M>(fill);
      }
    __builtin_amdgcn_s_setprio(0);
    PSTAMP(q7);
    PGCA_PBAR();
    PSTAMP(q8);
#ifdef PGCA_GEMM_TIMING2
    aL0 += q1 - q0; aB1 += q2 - q1; aM0 += q3 - q2; aB2 += q4 - q3;
    aL1 += q5 - q4; aB3 += q6 - q5; aM1 += q7 - q6; aB4 += q8 - q7;
#endif
  }
#ifdef PGCA_GEMM_TIMING2
  if (a.stat_sum && a.epilogue != PGCA_EPI_ROWSTATS && lane == 0) {
    float* o = a.stat_sum + ((size_t)blockIdx.x * 8 + wave) * 8;
    const float inv = 1.f / (float)nk;
    o[0] = aL0 * inv; o[1] = aB1 * inv; o[2] = aM0 * inv; o[3] = aB2 * inv;
    o[4] = aL1 * inv; o[5] = aB3 * inv; o[6] = aM1 * inv; o[7] = aB4 * inv;
  }
#endif
  if (PGCA_FILL_L || PGCA_FILL_M) {  // keep the fillers alive
    if (fill[0] + fill[1] + fill[2] + fill[3] + fill[4] + fill[5] + fill[6] + fill[7] == 12345.678f) acc[0][0][0][0] += 1.f;
  }
  if (wm == 0) PGCA_PBAR();  // re-join the groups
  __syncthreads();           // the epilogue stages through the same LDS

#ifdef PGCA_GEMM_TIMING
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts2)::"memory");
#endif
  run_epilogue(a, acc[0], smems, m0 + wm * 128, n0, tn, wn, lane, wave);
  run_epilogue(a, acc[1], smems, m0 + wm * 128 + 64, n0, tn, wn, lane, wave);
#ifdef PGCA_GEMM_TIMING
  asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts3)::"memory");
  if (a.stat_max && a.epilogue != PGCA_EPI_ROWSTATS && lane == 0) {
    float* o = a.stat_max + ((size_t)blockIdx.x * 8 + wave) * 8;
    o[0] = (float)(ts1 - ts0); o[1] = (float)(ts2 - ts1); o[2] = (float)(ts3 - ts2); o[3] = (float)(nk / 2);
    o[4] = (float)(ts0 & 0xFFFFFFFull); o[5] = (float)(ts3 & 0xFFFFFFFull);  // wrap at 2^28 clocks
    // in-kernel shader clock (MI355X_MICROARCH.md, DVFS item 6): (ts3 - ts0) / (tr3 - tr0) x 100 MHz
    unsigned long long tr3;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tr3)::"memory");
    o[6] = (float)(ts3 - ts0); o[7] = (float)(tr3 - tr0);
  }
#endif
}

constexpr size_t GEMM256S_LDS = (size_t)2 * PSTAGES * PTILE_BYTES;  // 128 KiB

template <int LA, int LB>
int launch_s(const pgca_gemm_args& a, int ntm, int ntn, int nkps, int nsplit, hipStream_t s) {
  static const bool attr_ok = hipFuncSetAttribute((const void*)gemm256s_kernel<LA, LB>,
                                                  hipFuncAttributeMaxDynamicSharedMemorySize,
                                                  (int)GEMM256S_LDS) == hipSuccess;  // once, thread-safe
  if (!attr_ok) {
    (void)hipGetLastError();
    set_error("gemm256s: cannot raise dynamic LDS limit");
    return PGCA_ERR_LAUNCH;
  }
  // stagger only where it can pay: >= 3 rounds of workgroups over the 256 CUs
  const int stagger = (ntm * ntn >= 3 * 256 && nsplit == 1) ? gemm_tuning().stagger : 0;
  hipLaunchKernelGGL((gemm256s_kernel<LA, LB>), dim3(ntm * ntn, nsplit), dim3(512), GEMM256S_LDS, s, a, ntm, ntn, nkps,
                     stagger);
  return check_launch("pgca_gemm_bf16(256 phase-staggered)");
}

}  // namespace

int pgca::launch_gemm256s(const pgca_gemm_args& a, int ntm, int ntn, int nk_per_split, int nsplit, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  switch (a.layout) {
    case PGCA_NT: return launch_s<0, 0>(a, ntm, ntn, nk_per_split, nsplit, s);
    case PGCA_NN: return launch_s<0, 1>(a, ntm, ntn, nk_per_split, nsplit, s);
    case PGCA_TN: return launch_s<1, 1>(a, ntm, ntn, nk_per_split, nsplit, s);
    default: set_error("pgca_gemm_bf16: unknown layout %d", a.layout); return PGCA_ERR_INVALID;
  }
}
