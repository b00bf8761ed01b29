// bf16 "NT" GEMM family for gfx950 (MI355X): C[m][n] = sum_k A[m][k] * B[n][k], fp32 accumulate, on the
// v_mfma_f32_16x16x32_bf16 matrix instruction.  Both operands are stored with K contiguous and a row pitch
// that is a multiple of 64 elements whose pad columns hold zeros, so K is always a whole number of 64-deep steps.
//
// Every GEMM-shaped hot op of CMA-ES is this one product with a different epilogue:
//   sampling      X = mean + sigma * Z * S^T          (Z bf16 Gaussian draws, S = C^{1/2} rounded to bf16)
//   rank-mu SYRK  G = Y^T Y  with Y^T stored [n][mu]  (upper-triangular tiles only, split-K slabs)
//   Newton-Schulz symmetric n x n products at fp32 accuracy from bf16 split planes
//                 (x = hi + mid + lo, six plane products), upper-triangular tiles only, mirrored on write
//
// Structure (one workgroup = WM x WN wavefronts of 64 lanes, BM x BN output tile, 64-deep K steps):
//   * operands go global -> LDS with global_load_lds_dwordx4 (no VGPR round trip).  One wave-instruction
//     moves 8 rows x 128 B; the LDS image is linear and the 16-byte chunks of a row are XOR-permuted on the
//     SOURCE address (chunk ^= row & 7), the fragment reads apply the same permutation, so the
//     ds_read_b128 of a 16-row fragment is free of bank conflicts;
//   * two or three LDS stages: the loads of the next one or two steps are in flight while a step is multiplied
//     (three stages: counted vmcnt and a bare s_barrier, so the loads stay outstanding across the barrier);
//   * the matrix instruction is fed B rows as its A operand and A rows as its B operand, so that a lane
//     ends up with four CONSECUTIVE output columns of one row and the epilogue stores 16 bytes per lane;
//   * blockIdx is remapped so that tiles sharing operand panels run on one XCD (one private L2).
#pragma once
#include "hip_common.h"

namespace evox {
namespace bg {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x4v = __attribute__((ext_vector_type(4))) float;
typedef __attribute__((address_space(3))) void* lds_vp;
typedef const __attribute__((address_space(1))) void* glb_vp;

constexpr int BK = 64;            // K elements per step
constexpr int ROW_BYTES = 128;    // LDS bytes per operand row and step
constexpr int MAXB = 2;           // products per launch (blockIdx.z)

__device__ __forceinline__ uint32_t f32_to_bf16_bits(float v) {
  const uint32_t u = __float_as_uint(v);
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float bf16_bits_to_f32(uint32_t b) { return __uint_as_float(b << 16); }

template <int NP>
struct Args {
  const uint16_t* a[MAXB][NP];   // planes of A (M rows), per product of the launch
  const uint16_t* b[MAXB][NP];   // planes of B (N rows)
  long long lda, ldb;            // row pitch in elements (multiple of 8)
  int M, N;
  int ksteps;                    // K / 64
  int ksteps_per_split;          // K steps handled by one workgroup
  int tiles_m, tiles_n;
  int ntiles, splits;            // grid = ntiles * splits * products, one dimension
  int batch;                     // products of the launch
  int probe;                     // timing experiments only: 1 = stage the first steps only, 2 = skip the multiply, 4 = skip the epilogue
};

// number of plane products for NP planes: pairs (pa, pb) with pa + pb <= NP - 1
template <int NP>
struct Terms;
template <>
struct Terms<1> {
  static constexpr int N = 1;
  static constexpr int PA[1] = {0};
  static constexpr int PB[1] = {0};
};
template <>
struct Terms<2> {
  static constexpr int N = 3;
  static constexpr int PA[3] = {0, 1, 0};
  static constexpr int PB[3] = {1, 0, 0};
};
template <>
struct Terms<3> {
  static constexpr int N = 6;
  static constexpr int PA[6] = {0, 2, 1, 0, 1, 0};     // smallest contributions first
  static constexpr int PB[6] = {2, 0, 1, 1, 0, 0};
};

template <int BM, int BN, int WM, int WN, int NP, int STAGES = 2>
struct Geometry {
  static constexpr int NW = WM * WN;
  static constexpr int NT = NW * 64;
  static constexpr int TM = BM / WM / 16;
  static constexpr int TN = BN / WN / 16;
  static constexpr int ROWS = NP * (BM + BN);
  static constexpr int STAGE_BYTES = ROWS * ROW_BYTES;
  static constexpr int NI = ROWS / 8;
  static constexpr int NJ = (NI + NW - 1) / NW;
  static constexpr int LDS_BYTES = STAGES * STAGE_BYTES;
  static_assert(STAGES >= 2 && STAGES <= 5, "two to five LDS stages");
  static_assert(LDS_BYTES <= 160 * 1024, "the stages must fit the 160 KiB of LDS");
  static_assert((STAGES - 2) * NJ <= 63, "vmcnt holds six bits");
  static_assert(BM % (WM * 16) == 0 && BN % (WN * 16) == 0, "wave tiles are multiples of 16");
  static_assert(BM % 8 == 0 && BN % 8 == 0, "planes start on 8-row boundaries");
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// A launch may carry SIDE workgroups behind the ones that multiply: a product of n = 1000 has 136 upper tiles of 64 x 64,
// so 120 of the 256 compute units would idle for its 20 us.  A `Side` functor with kEnabled = true is called by every
// workgroup whose index lies beyond the product's (side(index, count, lds)): small dependent-free jobs of the caller —
// matrix-vector products, the Gaussian draws of the next generation — then run in the product's shadow instead of as
// launches of their own.  The stage memory of such a workgroup is free for the functor's use.
struct NoSide {
  static constexpr bool kEnabled = false;
  __device__ __forceinline__ void operator()(int, int, char*) const {}
};

template <int BM, int BN, int WM, int WN, int NP, bool UPPER, int STAGES, class Epi, class Side = NoSide>
__global__ __launch_bounds__(WM* WN * 64) void gemm_nt_kernel(Args<NP> g, Epi epi, Side side) {
  using G = Geometry<BM, BN, WM, WN, NP, STAGES>;
  constexpr int NW = G::NW, TM = G::TM, TN = G::TN, NI = G::NI, NJ = G::NJ, STAGE_BYTES = G::STAGE_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  int nwg = (int)gridDim.x;
  if constexpr (Side::kEnabled) {
    nwg = g.ntiles * g.splits * g.batch;
    if ((int)blockIdx.x >= nwg) {
      side((int)blockIdx.x - nwg, (int)gridDim.x - nwg, smem);
      return;
    }
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // Workgroups b and b + 8 share an XCD (one private L2): the remap hands each XCD a contiguous range of
  // (product, K split, tile) triples with the tile index fastest, so neighbours in the range share operand rows.
  const int u = xcd_remap((int)blockIdx.x, nwg);
  const int z = u / (g.ntiles * g.splits);
  const int split = (u / g.ntiles) % g.splits;
  const int t = u % g.ntiles;
  int tm_i, tn_i;
  if constexpr (UPPER) {
    // row-major walk over the tiles (i, j) with i <= j of a tiles_n x tiles_n grid
    int i = 0, rem = t;
    while (rem >= g.tiles_n - i) {
      rem -= g.tiles_n - i;
      ++i;
    }
    tm_i = i;
    tn_i = i + rem;
  } else {
    tm_i = t / g.tiles_n;
    tn_i = t - tm_i * g.tiles_n;
  }
  const int m0 = tm_i * BM, n0 = tn_i * BN;
  const int ks0 = split * g.ksteps_per_split;
  const int nk = min(g.ksteps, ks0 + g.ksteps_per_split) - ks0;

  // per-lane source address of each of this wave's staging instructions (loop invariant)
  const char* src[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int i = j * NW + wave;                 // wave-instruction index inside a stage
    const int rbase = (i < NI ? i : 0) * 8;      // first of its 8 LDS rows (wave-uniform)
    const uint16_t* base;
    long long ld;
    int row, lim;
    if (rbase < NP * BM) {
      const int p = rbase / BM;
      base = g.a[z][p];
      ld = g.lda;
      row = m0 + (rbase - p * BM) + (lane >> 3);
      lim = g.M - 1;
    } else {
      const int q = rbase - NP * BM;
      const int p = q / BN;
      base = g.b[z][p];
      ld = g.ldb;
      row = n0 + (q - p * BN) + (lane >> 3);
      lim = g.N - 1;
    }
    row = min(row, lim);                         // rows past the edge re-read the last row (discarded later)
    const int chunk = (lane & 7) ^ ((lane >> 3) & 7);
    src[j] = reinterpret_cast<const char*>(base + (long long)row * ld) + chunk * 16 + (long long)ks0 * ROW_BYTES;
  }

  auto stage = [&](int buf, int kstep) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int i = j * NW + wave;
      if (i < NI)
        __builtin_amdgcn_global_load_lds((glb_vp)(src[j] + (long long)kstep * ROW_BYTES),
                                         (lds_vp)(smem + buf * STAGE_BYTES + i * 1024), 16, 0, 0);
    }
  };

  f32x4v acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) acc[a][b] = f32x4v{0.f, 0.f, 0.f, 0.f};

  const int wm0 = (wave / WN) * (TM * 16), wn0 = (wave % WN) * (TN * 16);
  // byte offset of this lane's 16-byte fragment piece inside a 16-row block, first and second half of the step
  const int lane_off0 = (lane & 15) * ROW_BYTES + (((lane >> 4) ^ (lane & 7)) << 4);
  const int lane_off1 = lane_off0 ^ 64;

  using T = Terms<NP>;
  // All fragment reads of a 64-deep step are issued before its first multiply (distinct registers), so the LDS
  // latency is paid once per step and the matrix instructions then issue back to back behind counted waits.
  auto compute = [&](int buf) {
    const char* sb = smem + buf * STAGE_BYTES;
    bf16x8 fa[2][NP][TM], fb[2][NP][TN];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int lo = ks ? lane_off1 : lane_off0;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
#pragma unroll
        for (int b = 0; b < TN; ++b)
          fb[ks][p][b] = *reinterpret_cast<const bf16x8*>(sb + (NP * BM + p * BN + wn0 + b * 16) * ROW_BYTES + lo);
#pragma unroll
        for (int a = 0; a < TM; ++a)
          fa[ks][p][a] = *reinterpret_cast<const bf16x8*>(sb + (p * BM + wm0 + a * 16) * ROW_BYTES + lo);
      }
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int q = 0; q < T::N; ++q)
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[ks][T::PB[q]][b], fa[ks][T::PA[q]][a], acc[a][b], 0, 0, 0);
    // scheduling groups: every LDS read of the step first, then the multiplies (waits stay counted per operand)
    __builtin_amdgcn_sched_group_barrier(0x100, 2 * NP * (TM + TN), 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 2 * T::N * TM * TN, 0);
  };

  if constexpr (STAGES == 2) {
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      if (kt + 1 < nk && !(g.probe & 1)) stage(cur ^ 1, kt + 1);
      if (!(g.probe & 2)) compute(cur);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  } else {
    // STAGES buffers, STAGES - 1 steps in flight.  Per step: wait until this wave's loads of step kt have landed
    // (the loads of the following steps may stay outstanding: a counted vmcnt), one barrier (every wave's share of
    // step kt is in LDS, and every wave has finished reading step kt - 1), refill the buffer of step kt - 1 with
    // step kt + STAGES - 1, multiply step kt.  The barrier is the bare instruction: a fenced one would drain the
    // loads in flight.
    constexpr int FULL = (NI % NW == 0) ? NJ : NJ - 1;       // loads per stage of the waves with one fewer
    const bool extra = (NI % NW != 0) && wave < NI % NW;     // this wave issues NJ per stage
    auto wait_behind = [&](int later) {                      // `later` younger stages may stay in flight
      if (extra) {
        if (later <= 0) wait_vmcnt<0>();
        else if (later == 1) wait_vmcnt<NJ>();
        else if (later == 2) wait_vmcnt<(STAGES > 3 ? 2 : 1) * NJ>();
        else wait_vmcnt<(STAGES > 4 ? 3 : 1) * NJ>();
      } else {
        if (later <= 0) wait_vmcnt<0>();
        else if (later == 1) wait_vmcnt<FULL>();
        else if (later == 2) wait_vmcnt<(STAGES > 3 ? 2 : 1) * FULL>();
        else wait_vmcnt<(STAGES > 4 ? 3 : 1) * FULL>();
      }
    };
#pragma unroll
    for (int s0 = 0; s0 < STAGES - 1; ++s0)
      if (s0 < nk) stage(s0, s0);
    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
      wait_behind(min(STAGES - 2, nk - 1 - kt));
      __builtin_amdgcn_s_barrier();
      if (kt + STAGES - 1 < nk && !(g.probe & 1)) stage(buf == 0 ? STAGES - 1 : buf - 1, kt + STAGES - 1);
      if (!(g.probe & 2)) compute(buf);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      buf = buf == STAGES - 1 ? 0 : buf + 1;
    }
  }

  const auto est = epi.init(z);
  const int mirror = UPPER ? (tm_i != tn_i ? 1 : 2) : 0;   // 1: tile above the diagonal, 2: on it
  if constexpr (Epi::kTileStaged && BM == BN) {
    // The whole BM x BN tile passes through LDS: the epilogue then stores complete rows — of the tile and, for a
    // symmetric product or a result that is kept in both orientations, of its transpose — instead of 16 rows of 64
    // bytes per wave-instruction and single bf16 elements for the mirrored half.
    constexpr int P = BN + 1;                            // odd pitch: column reads are free of bank conflicts
    static_assert(2 * BM * P * 4 <= G::LDS_BYTES, "tile and scratch reuse the stage memory");
    __syncthreads();                                     // every wave has finished reading the last stage
    float* tile = reinterpret_cast<float*>(smem);
    float* scratch = tile + BM * P;
    if (!(g.probe & 4)) {
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          float* q = tile + (wm0 + a * 16 + (lane & 15)) * P + wn0 + b * 16 + (lane >> 4) * 4;
          q[0] = acc[a][b][0]; q[1] = acc[a][b][1]; q[2] = acc[a][b][2]; q[3] = acc[a][b][3];
        }
      __syncthreads();
      epi.template store_tile<BM, P>(est, z, m0, n0, tile, scratch, mirror, (int)threadIdx.x, G::NT);
    }
  } else if constexpr (Epi::kRowStaged && (16 % TN == 0)) {
    // Row-contiguous stores: a lane holds 4 columns of one row and a wave-instruction would touch 16 rows with 64
    // bytes each.  Each wave passes its 16-row blocks through a private LDS patch and stores 16 / TN whole rows of
    // its TN * 64-byte wide tile per instruction instead.
    constexpr int W = TN * 16, PITCH = W + 4, RPI = 16 / TN;
    static_assert(NW * 16 * PITCH * 4 <= G::LDS_BYTES, "the epilogue patches reuse the stage memory");
    __syncthreads();                                   // every wave has finished reading the last stage
    float* wb = reinterpret_cast<float*>(smem) + wave * (16 * PITCH);
#pragma unroll
    for (int a = 0; a < TM; ++a) {
#pragma unroll
      for (int b = 0; b < TN; ++b)
        *reinterpret_cast<float4*>(wb + (lane & 15) * PITCH + b * 16 + (lane >> 4) * 4) =
            make_float4(acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]);
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const int row = i * RPI + lane / (TN * 4), col = (lane % (TN * 4)) * 4;
        const float4 v = *reinterpret_cast<const float4*>(wb + row * PITCH + col);
        if (!(g.probe & 4))
          epi(est, z, split, m0 + wm0 + a * 16 + row, n0 + wn0 + col, f32x4v{v.x, v.y, v.z, v.w}, mirror);
      }
    }
  } else {
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        const int m = m0 + wm0 + a * 16 + (lane & 15);
        const int n = n0 + wn0 + b * 16 + (lane >> 4) * 4;
        if (!(g.probe & 4)) epi(est, z, split, m, n, acc[a][b], mirror);
      }
  }
}

// ------------------------------------------------------------------------------------------------
// epilogues.  operator()(state, z, split, m, n, v, mirror): v[r] is element (m, n + r) of product z; mirror: 0 plain
// product, 1 tile above the diagonal of a symmetric product (write (n, m) too), 2 tile on its diagonal.
// ------------------------------------------------------------------------------------------------

// X[m][n] = mean[n] + sigma * v   (fp32 out, row pitch ldo a multiple of 4, base 16-byte aligned)
struct EpiSample {
  static constexpr bool kRowStaged = true;
  static constexpr bool kTileStaged = false;
  float* __restrict__ out;
  const float* __restrict__ mean;
  const float* __restrict__ sigma;
  long long ldo;
  int M, N;
  struct State { float s; };
  __device__ __forceinline__ State init(int) const { return State{sigma[0]}; }
  __device__ __forceinline__ void operator()(const State& st, int, int, int m, int n, f32x4v v, int) const {
    if (m >= M || n >= N) return;
    const float s = st.s;
    float* o = out + (long long)m * ldo + n;
    if (n + 3 < N) {
      const float4 mu = *reinterpret_cast<const float4*>(mean + n);
      *reinterpret_cast<float4*>(o) =
          make_float4(fmaf(s, v[0], mu.x), fmaf(s, v[1], mu.y), fmaf(s, v[2], mu.z), fmaf(s, v[3], mu.w));
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (n + r < N) o[r] = fmaf(s, v[r], mean[n + r]);
    }
  }
};

// raw partial sums of one K split: ws[split][m][n] (pitch ld, a multiple of 4; only computed tiles are written)
struct EpiSlab {
  static constexpr bool kRowStaged = true;
  static constexpr bool kTileStaged = false;
  float* __restrict__ ws;
  long long ld, slab;
  int M, N, splits;
  struct State {};
  __device__ __forceinline__ State init(int) const { return State{}; }
  __device__ __forceinline__ void operator()(const State&, int z, int split, int m, int n, f32x4v v, int) const {
    if (m >= M || n >= N) return;
    float* o = ws + ((long long)z * splits + split) * slab + (long long)m * ld + n;
    if (n + 3 < N) {
      *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (n + r < N) o[r] = v[r];
    }
  }
};

// One or two affine images of the product, each written as fp32 and / or as bf16 split planes, to (m, n) and, for
// the off-diagonal tiles of a symmetric product, to (n, m):
//     out_o = scale_o[z] * (alpha_o * v + beta_o * Cin_o[m][n] + gamma_o * [m == n])            o = 0, 1
// scale (optional) is read on the device.  keep (optional): keep[0] != 0 writes prev_o[z] in place of the result
// (a failed iteration passes the previous factors on), or nothing where prev_o[z] is null.
template <int NPO>
struct EpiAffine {
  static constexpr bool kRowStaged = false;
  static constexpr bool kTileStaged = true;        // square tiles go through store_tile, others through operator()
  struct Out {
    float* f32[MAXB];             // optional fp32 image, pitch ldf
    uint16_t* pl[MAXB][NPO];      // optional split planes, pitch ldp (pl[z][0] == nullptr: none)
    uint16_t* plt[MAXB][NPO];     // optional split planes of the TRANSPOSE (plain products only: the NT form takes the
                                  // right factor row-major, so a non-symmetric result that will be a right factor later
                                  // is stored both ways)
    const float* cin[MAXB];       // optional addend, pitch ldf
    const float* prev[MAXB];      // optional replacement under keep, pitch ldf
    const float* scale[MAXB];     // optional device scalar
    const float* beta_dev;        // optional device scalar multiplied into beta
    float alpha, beta, gamma;
  };
  Out o[2];
  int nout;
  long long ldf, ldp;
  int M, N;
  const float* keep;
  struct State {
    float sc[2];
    float bt[2];
    bool skip;
  };
  __device__ __forceinline__ State init(int z) const {
    State st{{1.f, 1.f}, {0.f, 0.f}, false};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (q >= nout) break;
      if (o[q].scale[z] != nullptr) st.sc[q] = o[q].scale[z][0];
      st.bt[q] = o[q].beta_dev != nullptr ? o[q].beta * o[q].beta_dev[0] : o[q].beta;
    }
    if (keep != nullptr) st.skip = keep[0] != 0.f;
    return st;
  }
  __device__ __forceinline__ void put(const Out& w, int z, int r, int c, float x) const {
    if (w.f32[z] != nullptr) w.f32[z][(long long)r * ldf + c] = x;
    if (w.pl[z][0] != nullptr) {
      float rem = x;
#pragma unroll
      for (int p = 0; p < NPO; ++p) {
        const uint32_t b = f32_to_bf16_bits(rem);
        w.pl[z][p][(long long)r * ldp + c] = (uint16_t)b;
        rem -= bf16_bits_to_f32(b);
      }
    }
  }
  // four consecutive elements of row r, columns c .. c + 3 (c a multiple of 4; r < M, c < N): fp32 image and planes
  __device__ __forceinline__ void put4(float* f32, uint16_t* const (&pl)[NPO], int r, int c, const float (&x)[4]) const {
    if (c + 3 < N && (ldf & 3) == 0) {
      if (f32 != nullptr) *reinterpret_cast<float4*>(f32 + (long long)r * ldf + c) = make_float4(x[0], x[1], x[2], x[3]);
      if (pl[0] != nullptr) {
        float rem[4] = {x[0], x[1], x[2], x[3]};
#pragma unroll
        for (int p = 0; p < NPO; ++p) {
          uint32_t b[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            b[k] = f32_to_bf16_bits(rem[k]);
            rem[k] -= bf16_bits_to_f32(b[k]);
          }
          *reinterpret_cast<uint2*>(pl[p] + (long long)r * ldp + c) = make_uint2(b[0] | (b[1] << 16), b[2] | (b[3] << 16));
        }
      }
      return;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (c + k >= N) break;
      if (f32 != nullptr) f32[(long long)r * ldf + c + k] = x[k];
      if (pl[0] != nullptr) {
        float rem = x[k];
#pragma unroll
        for (int p = 0; p < NPO; ++p) {
          const uint32_t b = f32_to_bf16_bits(rem);
          pl[p][(long long)r * ldp + c + k] = (uint16_t)b;
          rem -= bf16_bits_to_f32(b);
        }
      }
    }
  }

  // Epilogue of a square B x B tile held in LDS (`tile`, pitch P floats; `scratch`: a second array of that shape).
  // Every output is stored in complete rows; the transposed copy (mirror 1), the symmetric fill of a diagonal tile
  // (mirror 2) and the planes of the transpose (plt) read the finished values back from `scratch` column-wise.
  template <int B, int P>
  __device__ __forceinline__ void store_tile(const State& st, int z, int m0, int n0, const float* tile, float* scratch,
                                             int mirror, int tid, int nt) const {
    constexpr int QPR = B / 4, NQ = B * QPR;
    uint16_t* const none[NPO] = {};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (q >= nout) break;
      const Out& w = o[q];
      if (st.skip && w.prev[z] == nullptr) continue;
      const bool has_t = w.plt[z][0] != nullptr;
      const bool staged = mirror != 0 || has_t;
      for (int e = tid; e < NQ; e += nt) {
        const int row = e / QPR, c4 = (e - row * QPR) * 4;
        const int m = m0 + row, n = n0 + c4;
        float x[4] = {0.f, 0.f, 0.f, 0.f};
        if (m < M && n < N) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if (n + k >= N) break;
            float y;
            if (st.skip) {
              y = w.prev[z][(long long)m * ldf + n + k];
            } else {
              y = w.alpha * tile[row * P + c4 + k];
              if (w.cin[z] != nullptr) y = fmaf(st.bt[q], w.cin[z][(long long)m * ldf + n + k], y);
              if (m == n + k) y += w.gamma;
              y *= st.sc[q];
            }
            x[k] = y;
          }
          if (mirror != 2) put4(w.f32[z], w.pl[z], m, n, x);
        }
        if (staged) {
#pragma unroll
          for (int k = 0; k < 4; ++k) scratch[row * P + c4 + k] = x[k];
        }
      }
      if (!staged) continue;
      __syncthreads();
      for (int e = tid; e < NQ; e += nt) {
        const int row = e / QPR, c4 = (e - row * QPR) * 4;
        float x[4];
        if (mirror == 2) {
          // tile on the diagonal: both halves from the elements on and above it — symmetric to the last bit
          const int m = m0 + row, n = n0 + c4;
          if (m >= M || n >= N) continue;
#pragma unroll
          for (int k = 0; k < 4; ++k) x[k] = row <= c4 + k ? scratch[row * P + c4 + k] : scratch[(c4 + k) * P + row];
          put4(w.f32[z], w.pl[z], m, n, x);
          continue;
        }
        // row `row` of the transposed tile: element k is (original row c4 + k, original column row)
        const int m = n0 + row, n = m0 + c4;
        if (m >= N || n >= M) continue;
#pragma unroll
        for (int k = 0; k < 4; ++k) x[k] = scratch[(c4 + k) * P + row];
        if (mirror == 1) put4(w.f32[z], w.pl[z], m, n, x);
        if (has_t) put4(nullptr, w.plt[z], m, n, x);
      }
      __syncthreads();                                     // scratch is reused by the next output
    }
    (void)none;
  }

  __device__ __forceinline__ void operator()(const State& st, int z, int, int m, int n, f32x4v v, int mirror) const {
    if (m >= M || n >= N) return;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (q >= nout) break;
      const Out& w = o[q];
      if (st.skip && w.prev[z] == nullptr) continue;
      float x[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float y = 0.f;
        if (n + r < N) {
          if (st.skip) {
            y = w.prev[z][(long long)m * ldf + n + r];
          } else {
            y = w.alpha * v[r];
            if (w.cin[z] != nullptr) y = fmaf(st.bt[q], w.cin[z][(long long)m * ldf + n + r], y);
            if (m == n + r) y += w.gamma;
            y *= st.sc[q];
          }
        }
        x[r] = y;
      }
      if (mirror == 2) {
        // tile on the diagonal of a symmetric product: both halves come from the elements with m <= n, so the
        // result is symmetric to the last bit
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n + r < N && m <= n + r) {
            put(w, z, m, n + r, x[r]);
            if (m < n + r) put(w, z, n + r, m, x[r]);
          }
        continue;
      }
      if (n + 3 < N && (ldf & 3) == 0) {
        if (w.f32[z] != nullptr)
          *reinterpret_cast<float4*>(w.f32[z] + (long long)m * ldf + n) = make_float4(x[0], x[1], x[2], x[3]);
        if (w.pl[z][0] != nullptr) {
          float rem[4] = {x[0], x[1], x[2], x[3]};
#pragma unroll
          for (int p = 0; p < NPO; ++p) {
            uint32_t b[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              b[r] = f32_to_bf16_bits(rem[r]);
              rem[r] -= bf16_bits_to_f32(b[r]);
            }
            *reinterpret_cast<uint2*>(w.pl[z][p] + (long long)m * ldp + n) = make_uint2(b[0] | (b[1] << 16), b[2] | (b[3] << 16));
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n + r < N) put(w, z, m, n + r, x[r]);
      }
      if (w.plt[z][0] != nullptr) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (n + r >= N) break;
          float rem = x[r];
#pragma unroll
          for (int p = 0; p < NPO; ++p) {
            const uint32_t b = f32_to_bf16_bits(rem);
            w.plt[z][p][(long long)(n + r) * ldp + m] = (uint16_t)b;
            rem -= bf16_bits_to_f32(b);
          }
        }
      }
      if (mirror == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n + r < N) put(w, z, n + r, m, x[r]);
      }
    }
  }
};

// Largest dynamic LDS size is set once per instantiation (sizes above 64 KiB need the attribute).
template <int BM, int BN, int WM, int WN, int NP, bool UPPER, int STAGES, class Epi, class Side = NoSide>
int launch(Args<NP> g, const Epi& epi, int ntiles, int splits, int batch, hipStream_t st, const Side& side = Side{},
           int side_wgs = 0) {
  using G = Geometry<BM, BN, WM, WN, NP, STAGES>;
  auto kern = gemm_nt_kernel<BM, BN, WM, WN, NP, UPPER, STAGES, Epi, Side>;
  g.ntiles = ntiles;
  g.splits = splits;
  g.batch = batch;
  if (side_wgs < 0 || (side_wgs > 0 && !Side::kEnabled)) return (int)hipErrorInvalidValue;
  if (ntiles <= 0 || splits <= 0 || batch <= 0 || batch > MAXB || g.ksteps <= 0 ||
      (long long)(splits - 1) * g.ksteps_per_split >= g.ksteps || g.M <= 0 || g.N <= 0)
    return (int)hipErrorInvalidValue;
  static bool configured[16] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 16) return (int)hipErrorInvalidDevice;
  if (!configured[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       G::LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    configured[dev] = true;
  }
  kern<<<dim3(ntiles * splits * batch + side_wgs), G::NT, G::LDS_BYTES, st>>>(g, epi, side);
  return last_launch_status();
}

inline int upper_tiles(int t) { return t * (t + 1) / 2; }

}  // namespace bg
}  // namespace evox
