// "Duo" GEMM for gfx950: 256 x 128 output tile, FOUR waves (2 x 2, each 128 x 64), BK = 32, three LDS stages of
// (A 16 KiB | B 8 KiB) = 72 KiB, so TWO workgroups are resident per CU.  The 8-wave 256 x 256 kernel owns the whole CU:
// its prologue (first tile's HBM latency) and its epilogue (bias / GELU / residual read-modify-write, HBM-bound) run
// with the matrix pipe idle, which at K = 1024 is 30-45 % of the launch (tools/gemm_bench.py: 29 us of proj_fwd's
// 61 us, ~90 us of fc_fwd's 220 us).  With two independent workgroups per CU one computes while the other loads,
// stores or waits at its barrier.  Price: 1.5x the operand traffic per flop (L2 -> LDS) of the 256^2 tile.
//
// Pipeline per workgroup: LDS-DMA two K tiles ahead (inline asm, counted s_waitcnt vmcnt), one barrier per tile;
// fragment reads and MFMAs are left to the compiler's scheduler (the other workgroup covers the gaps).
#include "gemm_device.h"

namespace {

constexpr int DBM = 256, DBN = 128, DBK = 32, DSTAGES = 3;
constexpr int DA_BYTES = DBM * DBK * 2;            // 16 KiB
constexpr int DB_BYTES = DBN * DBK * 2;            // 8 KiB
constexpr int DSTAGE_BYTES = DA_BYTES + DB_BYTES;  // 24 KiB
constexpr size_t DUO_LDS = (size_t)DSTAGES * DSTAGE_BYTES;

__device__ __forceinline__ int swz4d(int q) { return (0x78 >> (2 * q)) & 3; }  // {0,2,3,1}

// One operand tile = NP 1-KiB pieces per wave (4 waves).  ROWS = tile extent along the non-K dimension.
//   KS = 0 ([ROWS][32 k], 64-B rows): piece = 16 rows; chunk c of row r stored at c ^ swz4((r >> 2) & 3).
//   KS = 1 ([32 k][ROWS cols]): ROWS = 256 -> 512-B k-rows, piece = 2 k-rows, 32-B slot s stored at s ^ h(k);
//                               ROWS = 128 -> 256-B k-rows, piece = 4 k-rows, same slot swizzle (8 slots).
template <int KS, int ROWS>
struct DmaD {
  static constexpr int NP = ROWS * DBK * 2 / 1024 / 4;  // pieces per wave: 4 (ROWS = 256) or 2 (ROWS = 128)
  unsigned goff[NP];
  __device__ __forceinline__ void init(int lane, int wave, int ld, int origin, int extent) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int j = wave * NP + i;
      if (KS == 0) {
        const int r = 16 * j + (lane >> 2);
        const int c = (lane & 3) ^ swz4d((lane >> 4) & 3);
        const int rg = min(origin + r, extent - 1) - origin;
        goff[i] = (unsigned)(rg * ld + c * 8) * 2u;
      } else if (ROWS == 256) {
        const int k = 2 * j + (lane >> 5);
        const int c16 = lane & 31;
        const int h = (k & 3) | (((k >> 3) & 1) << 2);
        const int col = (((c16 >> 1) ^ h) << 4) + ((c16 & 1) << 3);
        const int cg = min(origin + col, extent - 8) - origin;
        goff[i] = (unsigned)(k * ld + cg) * 2u;
      } else {
        const int k = 4 * j + (lane >> 4);
        const int c16 = lane & 15;
        const int h = (k & 3) | (((k >> 3) & 1) << 2);
        const int col = (((c16 >> 1) ^ h) << 4) + ((c16 & 1) << 3);
        const int cg = min(origin + col, extent - 8) - origin;
        goff[i] = (unsigned)(k * ld + cg) * 2u;
      }
    }
  }
  __device__ __forceinline__ void issue(const bf16_t* base, unsigned lds_tile, int wave) const {
    const unsigned long long b = (unsigned long long)base;
    u32x4 rs;
    rs[0] = (unsigned)b;
    rs[1] = (unsigned)(b >> 32) & 0xffffu;
    rs[2] = 0x7ffffff0u;
    rs[3] = 0x00020000u;
    const unsigned lds0 = lds_tile + (unsigned)wave * (NP * 1024u);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                   :
                   : "s"(lds0 + i * 1024u), "v"(goff[i]), "s"(rs)
                   : "memory");
    }
  }
};

template <int KS, int ROWS>
__device__ __forceinline__ bf16x8 read_frag_d(const unsigned char* lds, int wbase, int sub, int lane) {
  if (KS == 0) {
    const int row = wbase + sub * 16 + (lane & 15);
    const int pos = (lane >> 4) ^ swz4d((lane >> 2) & 3);
    return *reinterpret_cast<const bf16x8*>(lds + row * 64 + pos * 16);
  } else {
    return read_frag<1, ROWS * 2>(lds, wbase, sub, 0, lane);
  }
}

template <int N>
__device__ __forceinline__ void wait_vm_d() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int LA, int LB>
__global__ __launch_bounds__(256, 2) void gemm_duo_kernel(const pgca_gemm_args a, int ntm, int ntn, int nk_per_split) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smemd[];  // [3 stages][A 16 KiB | B 8 KiB]

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
  const int m0 = tm * DBM, n0 = tn * DBN;

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  const bf16_t* A = reinterpret_cast<const bf16_t*>(a.A);
  const bf16_t* B = reinterpret_cast<const bf16_t*>(a.B);

  DmaD<LA, DBM> da;
  DmaD<LB, DBN> db;
  da.init(lane, wave, a.lda, m0, LA == 0 ? a.M : ((a.M + 7) & ~7));
  db.init(lane, wave, a.ldb, n0, LB == 0 ? a.N : ((a.N + 7) & ~7));
  const bf16_t* abase = LA == 0 ? A + (size_t)m0 * a.lda : A + m0;
  const bf16_t* bbase = LB == 0 ? B + (size_t)n0 * a.ldb : B + n0;
  const size_t astep = LA == 0 ? (size_t)DBK : (size_t)DBK * a.lda;
  const size_t bstep = LB == 0 ? (size_t)DBK : (size_t)DBK * a.ldb;
  const unsigned lds0 = (unsigned)(size_t)LDS_PTR(smemd);

  f32x4 acc[2][4][4];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[h][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // split-K in units of 64-deep tiles (two ring tiles each) like the 256^2 kernel
  const int kt0 = blockIdx.y * nk_per_split * 2;
  const int nk = min(nk_per_split * 2, a.K / DBK - kt0);
  abase += (size_t)kt0 * astep;
  bbase += (size_t)kt0 * bstep;

  constexpr int PER_TILE = DmaD<LA, DBM>::NP + DmaD<LB, DBN>::NP;  // 6 DMA instructions per tile per wave
#pragma unroll
  for (int p = 0; p < DSTAGES - 1; ++p) {
    if (p < nk) {
      da.issue(abase + (size_t)p * astep, lds0 + p * DSTAGE_BYTES, wave);
      db.issue(bbase + (size_t)p * bstep, lds0 + p * DSTAGE_BYTES + DA_BYTES, wave);
    }
  }

  int st = 0;        // stage of tile kt
  int st_in = DSTAGES - 1;  // stage tile kt+2 goes to
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) wait_vm_d<PER_TILE>();  // tile kt landed; tile kt+1 may stay in flight
    else wait_vm_d<0>();
    __builtin_amdgcn_s_barrier();  // every wave's pieces of tile kt are visible; everyone is done reading tile kt-1
    if (kt + DSTAGES - 1 < nk) {
      da.issue(abase + (size_t)(kt + DSTAGES - 1) * astep, lds0 + st_in * DSTAGE_BYTES, wave);
      db.issue(bbase + (size_t)(kt + DSTAGES - 1) * bstep, lds0 + st_in * DSTAGE_BYTES + DA_BYTES, wave);
    }
    const unsigned char* la = smemd + st * DSTAGE_BYTES;
    const unsigned char* lb = la + DA_BYTES;
    bf16x8 fb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[j] = read_frag_d<LB, DBN>(lb, wn * 64, j, lane);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      bf16x8 fa[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = read_frag_d<LA, DBM>(la, wm * 128 + h * 64, i, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[h][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[h][i][j], 0, 0, 0);
    }
    st = st == DSTAGES - 1 ? 0 : st + 1;
    st_in = st_in == DSTAGES - 1 ? 0 : st_in + 1;
  }
  __syncthreads();  // the epilogue stages through the same LDS (4 waves x 8.5 KiB)

  run_epilogue_trunk(a, acc[0], smemd, m0 + wm * 128, n0, wn, lane, wave);
  run_epilogue_trunk(a, acc[1], smemd, m0 + wm * 128 + 64, n0, wn, lane, wave);
}

template <int LA, int LB>
int launch_duo(const pgca_gemm_args& a, int ntm, int ntn, int nkps, int nsplit, hipStream_t s) {
  static int attr_done = 0;
  if (!attr_done) {
    if (hipFuncSetAttribute((const void*)gemm_duo_kernel<LA, LB>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)DUO_LDS) != hipSuccess) {
      (void)hipGetLastError();
      set_error("gemm_duo: cannot raise dynamic LDS limit");
      return PGCA_ERR_LAUNCH;
    }
    attr_done = 1;
  }
  hipLaunchKernelGGL((gemm_duo_kernel<LA, LB>), dim3(ntm * ntn, nsplit), dim3(256), DUO_LDS, s, a, ntm, ntn, nkps);
  return check_launch("pgca_gemm_bf16(duo 256x128)");
}

}  // namespace

int pgca::launch_gemm_duo(const pgca_gemm_args& a, int nk_per_split, int nsplit, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!trunk_epilogue(a.epilogue)) return 1;  // caller falls back to the 256^2 kernel
  const int ncols = a.N;
  const int ntm = (a.M + DBM - 1) / DBM, ntn = (ncols + DBN - 1) / DBN;
  switch (a.layout) {
    case PGCA_NT: return launch_duo<0, 0>(a, ntm, ntn, nk_per_split, nsplit, s);
    case PGCA_NN: return launch_duo<0, 1>(a, ntm, ntn, nk_per_split, nsplit, s);
    case PGCA_TN: return launch_duo<1, 1>(a, ntm, ntn, nk_per_split, nsplit, s);
    default: set_error("pgca_gemm_bf16: unknown layout %d", a.layout); return PGCA_ERR_INVALID;
  }
}
