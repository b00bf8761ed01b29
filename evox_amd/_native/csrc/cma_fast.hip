// CMA-ES tell with the launch chain folded away (reference: es_variants/cma_es.py:137-196).
//
// The op-by-op tell of cma_step.hip issues ~24 kernels per generation at lambda = 10^4, n = 10^3; a dozen of them are
// 3-5 us helpers (partial-sum reductions, matrix-vector products, the path update, the probe residual, the guard, the
// control record) that each wait for the one before, and the 18 us of Gaussian draws of the next generation wait for
// all of them.  Here those jobs ride along with launches that leave compute units idle or are recomputed where they
// are consumed:
//
//   rank sort -> gather/transpose
//   -> rank-mu slabs            + side: mean shift from the gather's partial sums, whitening product C^{-1/2} step
//   -> covariance reduction     prologue: path / step-size update (every workgroup recomputes the n-vectors it needs);
//                               epilogue: the scaled covariance as bf16 split planes — the factorisation's scaling pass
//                               (the linear parts of the fifth-order polynomial are formed from C' in the products' epilogues)
//   -> A2 = Y0 Y0               + side: power-iteration product C' v (next call's scale), draws of the next generation
//   -> p  = (c0 I + c1 Y0) + A2 q + side: draws
//   -> S  = sqrt(s) Y0 p        + side: C'^{-1/2} probe, C'^{-1/2} v2 (condition estimate), draws
//   -> residual |S u - probe|   -> verdict, guard, control record, next call's scale, key split
//
// The scale s that maps the spectrum of C' into the polynomial's range is the estimate left by the PREVIOUS call
// (lambda_max moves by a fraction of a percent per generation and the estimate carries 2 % head room), which is what
// lets the covariance reduction emit the scaled planes directly; the estimate for the next call is refreshed from C'
// in the shadow of the first product.  Eleven launches instead of twenty-four, and the sampling product of the next
// ask finds its draws and its keys ready.
#include "bf16_gemm.h"
#include "cma_step.h"
#include "ns_scal.h"
#include "philox.h"

extern "C" {
int evox_hip_rank_sort_scalar(const float* keys, int n, int64_t* order, int64_t* rank_out, void* stream);
int evox_hip_cma_gather_t(const float* pop, int64_t ldpop, const int64_t* order, const float* mean, const float* rowscale,
                          const float* sigma, void* yt, int64_t ldy, float* colsum, int rb, float* step, int mu, int n,
                          int lam, void* stream);
int evox_hip_bf16_cov_splits(int n, int mu);
}

using namespace evox;
using namespace evox::bg;
using namespace evox::nsk;

namespace {

// ------------------------------------------------------------------------------------------------------------------
// side jobs
// ------------------------------------------------------------------------------------------------------------------
struct SideMatvec {
  const float* A;        // n x n row-major; nullptr: no job
  const float* x;        // n floats; nullptr (and no xparts): the probe vector
  const float* xparts;   // rb x n partial vectors whose sum (fixed order) is x, or nullptr
  float* x_out;          // the summed x is stored here by the job's first workgroup (with xparts)
  float* y;              // n floats out
  float* stats;          // scalar block: |y|^2 partials into row 0 (and, with norms, max row-abs-sum / trace partials
                         // into rows kRowInf / kRowDiag), one entry per workgroup of the job; or nullptr
  int rb, norms, wgs;
};

struct FastSide {
  static constexpr bool kEnabled = true;
  int n;
  SideMatvec mv[2];
  // Gaussian draws of the next generation: rows [row0, row1) of its buffer, element (row_offset + r) * n + c of the draw
  // stream of split(key)[1].  keys_out (optional): the split itself, written by the first drawing workgroup.
  const int32_t* key;
  uint16_t* z;
  long long ldz, row_offset;
  int row0, row1;
  int32_t* keys_out;

  __device__ void matvec(const SideMatvec& j, int part, char* smem) const {
    float* xs = reinterpret_cast<float*>(smem);
    float* red = xs + ((n + 63) & ~63);
    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
    if (j.xparts != nullptr && (n & 3) == 0) {
      // x = sum of rb partial vectors: column quads over 256 threads, the partials dealt over the nt / 256 thread groups
      // (their loads unrolled: a side workgroup is alone on its compute unit), groups added in a fixed order
      const int G = nt >> 8, grp = tid >> 8, nq = n >> 2;
      float4* pbuf = reinterpret_cast<float4*>(xs + ((n + 63) & ~63) + 64);
      for (int q = tid & 255; q < nq; q += 256) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
        for (int b = grp; b < j.rb; b += G) {
          const float4 v = *reinterpret_cast<const float4*>(j.xparts + (long long)b * n + 4 * q);
          acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        pbuf[grp * nq + q] = acc;
      }
      __syncthreads();
      for (int q = tid; q < nq; q += nt) {
        float4 acc = pbuf[q];
        for (int g = 1; g < G; ++g) {
          const float4 v = pbuf[g * nq + q];
          acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        reinterpret_cast<float4*>(xs)[q] = acc;
      }
    } else {
      for (int c = tid; c < n; c += nt) {
        float v;
        if (j.xparts != nullptr) {
          float p[4] = {0.f, 0.f, 0.f, 0.f};
          for (int b = 0; b < j.rb; ++b) p[b & 3] += j.xparts[(long long)b * n + c];
          v = (p[0] + p[1]) + (p[2] + p[3]);
        } else {
          v = j.x != nullptr ? j.x[c] : probe_v(c);
        }
        xs[c] = v;
      }
    }
    __syncthreads();
    if (j.xparts != nullptr && part == 0 && j.x_out != nullptr)
      for (int c = tid; c < n; c += nt) j.x_out[c] = xs[c];
    const int per = (n + j.wgs - 1) / j.wgs;
    const int r0 = part * per, r1 = min(n, r0 + per);
    float sq = 0.f, mx = 0.f, dg = 0.f;
    if ((n & 3) == 0) {
      // R rows per wavefront at a time, every load of the R rows issued before the first use: the few workgroups of a
      // side job have to keep enough bytes in flight themselves (n <= 1024: at most four 16-byte loads per lane and row)
      constexpr int R = 4;
      const int nq = n >> 2;
      const float4* x4 = reinterpret_cast<const float4*>(xs);
      for (int rb0 = r0 + wave * R; rb0 < r1; rb0 += nw * R) {
        float acc[R], asum[R];
#pragma unroll
        for (int k = 0; k < R; ++k) acc[k] = asum[k] = 0.f;
        for (int q0 = 0; q0 < nq; q0 += 256) {
          float4 a[R][4];
#pragma unroll
          for (int k = 0; k < R; ++k) {
            const int row = min(rb0 + k, r1 - 1);
            const float4* r4 = reinterpret_cast<const float4*>(j.A + (size_t)row * n);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int q = q0 + u * 64 + lane;
              a[k][u] = q < nq ? r4[q] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int q = q0 + u * 64 + lane;
            const float4 v = q < nq ? x4[q] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int k = 0; k < R; ++k) {
              acc[k] += (a[k][u].x * v.x + a[k][u].y * v.y) + (a[k][u].z * v.z + a[k][u].w * v.w);
              asum[k] += (fabsf(a[k][u].x) + fabsf(a[k][u].y)) + (fabsf(a[k][u].z) + fabsf(a[k][u].w));
            }
          }
        }
#pragma unroll
        for (int k = 0; k < R; ++k) {
          const int row = rb0 + k;
          const float t = wave_sum(acc[k]);
          const float am = j.norms ? wave_sum(asum[k]) : 0.f;
          if (row < r1) {
            if (lane == 0) j.y[row] = t;
            sq = fmaf(t, t, sq);
            if (j.norms) {
              mx = fmaxf(mx, am);
              dg += j.A[(size_t)row * n + row];
            }
          }
        }
      }
    } else {
      for (int row = r0 + wave; row < r1; row += nw) {
        const float* r = j.A + (size_t)row * n;
        float acc = 0.f, asum = 0.f;
        for (int q = lane; q < n; q += 64) {
          acc = fmaf(r[q], xs[q], acc);
          asum += fabsf(r[q]);
        }
        acc = wave_sum(acc);
        if (lane == 0) j.y[row] = acc;
        sq = fmaf(acc, acc, sq);
        if (j.norms) {
          mx = fmaxf(mx, wave_sum(asum));
          dg += r[row];
        }
      }
    }
    if (j.stats == nullptr) return;
    if (lane == 0) {
      red[wave] = sq;
      red[16 + wave] = mx;
      red[32 + wave] = dg;
    }
    __syncthreads();
    if (tid == 0) {
      float s = 0.f, m = 0.f, d = 0.f;
      for (int w = 0; w < nw; ++w) {
        s += red[w];
        m = fmaxf(m, red[16 + w]);
        d += red[32 + w];
      }
      part_row(j.stats, 0)[part] = s;
      if (j.norms) {
        part_row(j.stats, kRowInf)[part] = m;
        part_row(j.stats, kRowDiag)[part] = d;
      }
    }
  }

  __device__ void draws(int part, int parts) const {
    const uint32_t k0 = (uint32_t)key[0], k1 = (uint32_t)key[1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    if (part == 0 && tid < 2 && keys_out != nullptr) {
      const u32x4 b = philox_block(k0, k1, (uint64_t)tid, kStreamSplit);
      keys_out[2 * tid] = (int32_t)b.x;
      keys_out[2 * tid + 1] = (int32_t)b.y;
    }
    const u32x4 sk = philox_block(k0, k1, 1ull, kStreamSplit);      // split(key)[1]: the sampling key
    const int bpr = n >> 2;
    for (int r = row0 + part * nw + wave; r < row1; r += parts * nw) {
      const uint64_t b0 = (uint64_t)(row_offset + r) * (uint64_t)bpr;
      uint16_t* orow = z + (long long)r * ldz;
      // two counter blocks per lane and iteration: independent chains for the few wavefronts a side workgroup has
      for (int cb = lane; cb < bpr; cb += 128) {
        const int cb2 = cb + 64;
        const u32x4 ba = philox_block(sk.x, sk.y, b0 + (uint64_t)cb, kStreamDraw);
        const u32x4 bb = philox_block(sk.x, sk.y, b0 + (uint64_t)(cb2 < bpr ? cb2 : cb), kStreamDraw);
        const f32x4 ga = normal4(ba), gb = normal4(bb);
        *reinterpret_cast<uint2*>(orow + 4 * cb) = make_uint2(f32_to_bf16_bits(ga.x) | (f32_to_bf16_bits(ga.y) << 16),
                                                              f32_to_bf16_bits(ga.z) | (f32_to_bf16_bits(ga.w) << 16));
        if (cb2 < bpr)
          *reinterpret_cast<uint2*>(orow + 4 * cb2) = make_uint2(f32_to_bf16_bits(gb.x) | (f32_to_bf16_bits(gb.y) << 16),
                                                                 f32_to_bf16_bits(gb.z) | (f32_to_bf16_bits(gb.w) << 16));
      }
    }
  }

  __device__ void operator()(int sid, int nside, char* smem) const {
    int base = 0;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (mv[q].A == nullptr) continue;
      if (sid < base + mv[q].wgs) {
        matvec(mv[q], sid - base, smem);
        return;
      }
      base += mv[q].wgs;
    }
    if (key != nullptr && nside > base && row1 > row0) draws(sid - base, nside - base);
  }
};

// ------------------------------------------------------------------------------------------------------------------
// covariance reduction + path update + scaling pass of the factorisation
// ------------------------------------------------------------------------------------------------------------------
struct PathConst { float cs, cc, mueff, cm, damps, chiN, c1, cmu; };

struct CovFast {
  const float* ws;          // K-split slabs of the rank-mu product
  int splits;
  const float* packed;      // or: the rank-mu term summed already, as tile-packed upper triangle (row-sharded generation)
  const float* poison;      // optional, one float: positive = some process dropped selected rows, the result becomes NaN
  const float *C, *mean, *step, *whitened, *ps, *pc, *sigma, *iter;
  const float* prev_rec;    // record of the previous factorisation: [4] = its scale estimate for this call
  PathConst k;
  float *C_o, *mean_o, *ps_o, *pc_o, *scalars;
  uint16_t* y0;             // two bf16 planes of C' / s, pitch ldp, plane stride `plane`
  long long ldp, plane;
  float* scal;              // [8] 1/s, [11] 1/sqrt(s), [12] sqrt(s), [13] s
  int n, tiles;
};

// NT threads per workgroup (one 64 x 64 tile each): 1024 keep four times the loads of a tile in flight — the kernel has
// only T (T + 1) / 2 workgroups and is bound by latency, not by bytes (12.9 -> 11.4 us; dealing every tile above the
// diagonal to two workgroups, 256 in all at n = 1000, gained nothing more).
template <int MAXS, int NT>
__global__ __launch_bounds__(NT) void cov_fast_kernel(CovFast a) {
  __shared__ float tr[64][65];
  __shared__ float red[NT / 64];
  const int n = a.n;
  // ---- path / step-size update: every workgroup derives the scalars and the entries of pc' it needs ----
  const float sg = a.sigma[0];
  const float ca = sqrtf(a.k.cs * (2.f - a.k.cs) * a.k.mueff) * a.k.cm / sg;
  float sq = 0.f;
  // (every group of 256 threads sums the same four strided slots: the value does not depend on NT)
  for (int c = threadIdx.x & 255; c < n; c += 256) {
    const float p = (1.f - a.k.cs) * a.ps[c] + ca * a.whitened[c];
    if (blockIdx.x == 0 && threadIdx.x < 256) a.ps_o[c] = p;
    sq = fmaf(p, p, sq);
  }
  sq = wave_sum(sq);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sq;
  __syncthreads();
  const float norm = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
  const float denom = sqrtf(1.f - powf(1.f - a.k.cs, 2.f * a.iter[0]));
  const float hsig = (norm / denom < (1.4f + 2.f / (float)(n + 1)) * a.k.chiN) ? 1.f : 0.f;
  const float cb = hsig * sqrtf(a.k.cc * (2.f - a.k.cc) * a.k.mueff) * a.k.cm / sg;
  const float alpha = (1.f - a.k.c1 - a.k.cmu) + a.k.c1 * (1.f - hsig) * a.k.cc * (2.f - a.k.cc);
  const float s_in = a.prev_rec[4];
  const float s = (s_in > 0.f && s_in < 3.0e38f) ? s_in : 1.f;
  const float is = 1.f / s;
  const bool bad = a.poison != nullptr && a.poison[0] > 0.f;
  if (blockIdx.x == 0) {
    const float sigma_new = sg * expf((a.k.cs / a.k.damps) * (norm / a.k.chiN - 1.f));
    for (int c = threadIdx.x; c < n; c += NT) {
      const float st = a.step[c];
      a.mean_o[c] = fmaf(a.k.cm, st, a.mean[c]);
      a.pc_o[c] = (1.f - a.k.cc) * a.pc[c] + cb * st;
      a.scalars[4 + c] = sigma_new;
    }
    if (threadIdx.x == 0) {
      a.scalars[0] = sigma_new;
      a.scalars[1] = alpha;
      a.scalars[2] = hsig;
      a.scalars[3] = norm;
      a.scal[13] = s;
      a.scal[12] = sqrtf(s);
      a.scal[11] = 1.f / sqrtf(s);
      a.scal[8] = is;
    }
  }
  auto pc_new = [&](int c) { return c < n ? (1.f - a.k.cc) * a.pc[c] + cb * a.step[c] : 0.f; };

  // ---- C' = alpha C + c1 pc' pc'^T + cmu sum of slabs on the upper 64 x 64 tiles, mirrored; the planes of Y0 = C' / s alongside ----
  int i = 0, rem = blockIdx.x;
  while (rem >= a.tiles - i) {
    rem -= a.tiles - i;
    ++i;
  }
  const int j = i + rem;
  const int r0 = i * 64, c0 = j * 64;
  const int q = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const long long slab = (long long)n * n;
  const int cc = c0 + 4 * q;
  float pcc[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) pcc[k] = pc_new(cc + k);
  auto emit = [&](int r, int c, float v) {
    const long long e = (long long)r * n + c;
    const float y = v * is;
    a.C_o[e] = v;
    const uint32_t b0 = f32_to_bf16_bits(y);
    const uint32_t b1 = f32_to_bf16_bits(y - bf16_bits_to_f32(b0));
    a.y0[(long long)r * a.ldp + c] = (uint16_t)b0;
    a.y0[a.plane + (long long)r * a.ldp + c] = (uint16_t)b1;
  };
  constexpr int RG = NT / 16;                       // row groups handled at once
#pragma unroll
  for (int p = 0; p < 64 / RG; ++p) {
    const int rl = rg + RG * p, r = r0 + rl;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (r < n && cc < n) {                          // n is a multiple of 4: the quad is inside or outside as a whole
      const long long e = (long long)r * n + cc;
      float4 sum;
      if (a.packed != nullptr) {
        sum = *reinterpret_cast<const float4*>(a.packed + (long long)blockIdx.x * 4096 + rl * 64 + 4 * q);
      } else {
        float4 g[MAXS];
#pragma unroll
        for (int t = 0; t < MAXS; ++t)
          g[t] = t < a.splits ? *reinterpret_cast<const float4*>(a.ws + t * slab + e) : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int t = MAXS; t < a.splits; ++t) {
          const float4 u = *reinterpret_cast<const float4*>(a.ws + t * slab + e);
          g[0].x += u.x; g[0].y += u.y; g[0].z += u.z; g[0].w += u.w;
        }
        sum = g[0];
#pragma unroll
        for (int t = 1; t < MAXS; ++t) { sum.x += g[t].x; sum.y += g[t].y; sum.z += g[t].z; sum.w += g[t].w; }
      }
      const float4 cv = *reinterpret_cast<const float4*>(a.C + e);
      const float pr = pc_new(r);
      v[0] = fmaf(alpha, cv.x, fmaf(a.k.c1, pr * pcc[0], a.k.cmu * sum.x));
      v[1] = fmaf(alpha, cv.y, fmaf(a.k.c1, pr * pcc[1], a.k.cmu * sum.y));
      v[2] = fmaf(alpha, cv.z, fmaf(a.k.c1, pr * pcc[2], a.k.cmu * sum.z));
      v[3] = fmaf(alpha, cv.w, fmaf(a.k.c1, pr * pcc[3], a.k.cmu * sum.w));
      if (bad) v[0] = v[1] = v[2] = v[3] = __builtin_nanf("");
      if (i != j) {
        float y[4];
        uint32_t b0[4], b1[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          y[k] = v[k] * is;
          b0[k] = f32_to_bf16_bits(y[k]);
          b1[k] = f32_to_bf16_bits(y[k] - bf16_bits_to_f32(b0[k]));
        }
        *reinterpret_cast<float4*>(a.C_o + e) = make_float4(v[0], v[1], v[2], v[3]);
        uint16_t* pl = a.y0 + (long long)r * a.ldp + cc;
        *reinterpret_cast<uint2*>(pl) = make_uint2(b0[0] | (b0[1] << 16), b0[2] | (b0[3] << 16));
        *reinterpret_cast<uint2*>(pl + a.plane) = make_uint2(b1[0] | (b1[1] << 16), b1[2] | (b1[3] << 16));
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) tr[rl][4 * q + k] = v[k];
  }
  __syncthreads();
  const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
  constexpr int NG = NT / 64;
  if (i == j) {
    // a diagonal tile takes both halves from its upper triangle: the result is symmetric to the last bit
#pragma unroll 4
    for (int rr = 0; rr < 64 / NG; ++rr) {
      const int rl = rr * NG + grp, r = r0 + rl;
      if (r < n && c0 + col < n) emit(r, c0 + col, rl <= col ? tr[rl][col] : tr[col][rl]);
    }
    return;
  }
#pragma unroll 4
  for (int rr = 0; rr < 64 / NG; ++rr) {
    const int rl = rr * NG + grp;                  // row of the mirrored tile = column of the computed one
    const int r = c0 + rl, c = r0 + col;
    if (r < n && c < n) emit(r, c, tr[col][rl]);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// residual of the finished factorisation and the closing bookkeeping
// ------------------------------------------------------------------------------------------------------------------
// |S u - probe|^2 and |probe|^2 per block, u = C'^{-1/2} probe from the side job of the last product: for E = S Z - I
// and a random direction, E|E v|^2 = |E|_F^2 / n.
__global__ __launch_bounds__(256) void residual_kernel(const float* __restrict__ S, int n, const float* __restrict__ u,
                                                       float* scal) {
  __shared__ float sd[4], sv[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float d2 = 0.f, v2 = 0.f;
  for (int row = blockIdx.x * 4 + wave; row < n; row += gridDim.x * 4) {
    const float* r = S + (size_t)row * n;
    float acc = 0.f;
    if ((n & 3) == 0) {
      const float4* r4 = reinterpret_cast<const float4*>(r);
      const float4* u4 = reinterpret_cast<const float4*>(u);
      for (int q = lane; q < (n >> 2); q += 64) {
        const float4 a = r4[q], v = u4[q];
        acc += (a.x * v.x + a.y * v.y) + (a.z * v.z + a.w * v.w);
      }
    } else {
      for (int q = lane; q < n; q += 64) acc = fmaf(r[q], u[q], acc);
    }
    acc = wave_sum(acc);
    const float v = probe_v(row), d = acc - v;
    d2 = fmaf(d, d, d2);
    v2 = fmaf(v, v, v2);
  }
  if (lane == 0) { sd[wave] = d2; sv[wave] = v2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part_row(scal, kRowRes)[blockIdx.x] = (sd[0] + sd[1]) + (sd[2] + sd[3]);
    part_row(scal, kMaxPow - 1)[blockIdx.x] = (sv[0] + sv[1]) + (sv[2] + sv[3]);
  }
}

struct Finish {
  float* scal;              // block of this call
  const float* scal2;       // partials of the condition product (C'^{-1/2} v2), or unused
  int res_blocks, pw_wgs, cond_wgs, cond, n;
  long long ldp;
  const float *prev_sqrt, *prev_inv;
  float *out_sqrt, *out_inv;
  uint16_t* out_b16;
  const float* w;           // C' v (power product of this call)
  const float* w2;          // C'^{-1/2} v2 (cond only)
  const float* ctrl_vec;    // 2 x n vectors of the incoming control
  const float* prev_rec;
  float *ctrl_vec_o, *rec_o, *rec_host;
  const float* count_in;    // generation counter and its successor (optional)
  float* count_out;
};

// Verdict from the residual partials (every block derives it: only a failed factorisation copies the previous factors
// over the outputs), then by block 0: the scale estimate for the next call from the power product, the condition
// estimates (refreshed or carried over), the normalised control vectors and the 7-float record.
__global__ __launch_bounds__(256) void finish_kernel(Finish f) {
  __shared__ float red[4];
  const int n = f.n;
  const float d2 = block_reduce(part_row(f.scal, kRowRes), f.res_blocks, false, red);
  const float v2 = block_reduce(part_row(f.scal, kMaxPow - 1), f.res_blocks, false, red);
  const float res = sqrtf(d2 / fmaxf(v2, 1e-38f));
  const bool failed = !(res < 0.25f) && f.prev_sqrt != nullptr && f.prev_inv != nullptr;
  if (blockIdx.x == 0) {
    const float inf_norm = block_reduce(part_row(f.scal, kRowInf), f.pw_wgs, true, red);
    const float trace = block_reduce(part_row(f.scal, kRowDiag), f.pw_wgs, false, red);
    const float last = block_reduce(part_row(f.scal, 0), f.pw_wgs, false, red);
    const float safe_s = fmaxf(trace / (float)n, inf_norm / 1.9f);
    const float lam = sqrtf(last) * 1.02f;
    const float s_next = (lam > 0.f && lam < 3.0e38f) ? fminf(lam, safe_s * 1.9f) : safe_s;
    const float inv = rsqrtf(fmaxf(last, 1e-38f));
    for (int c = threadIdx.x; c < n; c += 256) f.ctrl_vec_o[c] = f.w[c] * inv;
    float c_lo = f.prev_rec[0], c_hi = f.prev_rec[1];
    if (f.cond) {
      const float inf2 = block_reduce(part_row(f.scal2, kRowInf), f.cond_wgs, true, red);
      const float sq2 = block_reduce(part_row(f.scal2, 0), f.cond_wgs, false, red);
      c_lo = s_next * sq2;
      c_hi = s_next * inf2 * inf2;
      const float inv2 = rsqrtf(fmaxf(sq2, 1e-38f));
      for (int c = threadIdx.x; c < n; c += 256) f.ctrl_vec_o[n + c] = f.w2[c] * inv2;
    } else {
      for (int c = threadIdx.x; c < n; c += 256) f.ctrl_vec_o[n + c] = f.ctrl_vec[n + c];
    }
    if (threadIdx.x == 0) {
      const float rec[7] = {c_lo, c_hi, 1.f / sqrtf(s_next), sqrtf(s_next), s_next, res, failed ? 1.f : 0.f};
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        f.rec_o[k] = rec[k];
        f.scal[9 + k] = rec[k];
        if (f.rec_host != nullptr) f.rec_host[k] = rec[k];
      }
      if (f.count_out != nullptr) *f.count_out = *f.count_in + 1.0f;
    }
  }
  if (!failed) return;
  const long long total = (long long)n * n;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const float sv = f.prev_sqrt[e];
    f.out_sqrt[e] = sv;
    f.out_inv[e] = f.prev_inv[e];
    if (f.out_b16 != nullptr) f.out_b16[(e / n) * f.ldp + (e % n)] = (uint16_t)f32_to_bf16_bits(sv);
  }
}

// Side jobs on their own (row-sharded generation: the whitening product has no rank-mu launch to ride along with).
__global__ __launch_bounds__(1024) void side_only_kernel(FastSide side) {
  extern __shared__ __attribute__((aligned(16))) char side_smem[];
  side((int)blockIdx.x, (int)gridDim.x, side_smem);
}

// ------------------------------------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------------------------------------
template <int NPO>
EpiAffine<NPO> affine(int n, long long ldp) {
  EpiAffine<NPO> e{};
  e.nout = 1;
  e.ldf = n;
  e.ldp = ldp;
  e.M = e.N = n;
  e.keep = nullptr;
  return e;
}

// one symmetric plane product (upper 64 x 64 tiles, two planes per operand) with side workgroups behind it
template <int WM, int WN, int NPO>
int product(const uint16_t* A, const uint16_t* B, long long plane, long long ldp, int n, const EpiAffine<NPO>& epi,
            const FastSide& side, int side_wgs, hipStream_t st) {
  Args<2> g{};
  for (int p = 0; p < 2; ++p) {
    g.a[0][p] = g.a[1][p] = A + p * plane;
    g.b[0][p] = g.b[1][p] = B + p * plane;
  }
  g.lda = g.ldb = ldp;
  g.M = g.N = n;
  g.ksteps = (n + 63) / 64;
  g.ksteps_per_split = g.ksteps;
  const int T = (n + 63) / 64;
  g.tiles_m = g.tiles_n = T;
  return launch<64, 64, WM, WN, 2, true, 3, EpiAffine<NPO>, FastSide>(g, epi, upper_tiles(T), 1, 1, st, side, side_wgs);
}

inline int ksteps_of(long long k) { return (int)((k + 63) / 64); }

}  // namespace

// Extra arguments of the folded tell.
struct CmaFast {
  const int32_t* key;        // key of the state (the next ask splits it), or nullptr: no draws ahead
  void* z_next;              // rows x ldn bf16: draw buffer of the next generation
  int32_t* keys_next;        // [2][2]: split(key)
  const float* count_in;     // generation counter on the device and its successor for the next ask (optional)
  float* count_next;
  int64_t row_offset;
  int rows;                  // rows to draw (lam)
  int wide;                  // 1: sixteen wavefronts per product workgroup (4 x 4), 0: eight (2 x 4)
};

template <int WM, int WN>
static int tell_fast(const CmaTell* t, const CmaFast* f, hipStream_t st) {
  const int n = t->n;
  const long long nn = (long long)n * n;
  int e = 0;
  int splits = 0;
  if (t->sharded) {
    // the selected rows were reduced on every process and summed already: only the whitening product of the summed
    // mean shift is left of the front part
    FastSide side{};
    side.n = n;
    const int wgs = 48;
    side.mv[0] = SideMatvec{t->invsqrtC, t->step, nullptr, nullptr, t->whitened, nullptr, 0, 0, wgs};
    const size_t smem = (size_t)(((n + 63) & ~63) + 128) * sizeof(float);
    side_only_kernel<<<wgs, 1024, smem, st>>>(side);
    e = last_launch_status();
    if (e) return e;
  } else {
  e = evox_hip_rank_sort_scalar(t->fitness, t->lam, t->order, nullptr, st);
  if (e) return e;
  e = evox_hip_cma_gather_t(t->pop, t->ldpop, t->order, t->mean, t->rowscale, t->sigma, t->yt, t->ldmu, t->colsum, t->rb,
                            nullptr, t->mu, n, t->lam, st);
  if (e) return e;

  // ---- rank-mu slabs; side: mean shift and whitening product ----
  // K splits of the rank-mu product: as many as keep (upper 128 x 128 tiles) x splits within the compute units that the
  // side job (mean shift + whitening product: it wants three dozen of them) leaves
  const int ks = ksteps_of(t->mu);
  const int T128 = (n + 127) / 128;
  splits = (kNumCU - 36) / upper_tiles(T128);
  {
    const int most = evox_hip_bf16_cov_splits(n, t->mu);       // what the work space is sized for
    splits = splits > most ? most : (splits < 1 ? 1 : splits);
  }
  int per = (ks + splits - 1) / splits;
  splits = (ks + per - 1) / per;
  if (t->ldmu < (long long)ks * 64 || (t->ldmu & 7) || ((uintptr_t)t->yt & 15) || ((uintptr_t)t->cov_ws & 15) ||
      t->cov_ws_floats < (long long)splits * nn)
    return (int)hipErrorInvalidValue;
  {
    const int T = (n + 127) / 128;
    Args<1> g{};
    g.a[0][0] = g.b[0][0] = (const uint16_t*)t->yt;
    g.lda = g.ldb = t->ldmu;
    g.M = g.N = n;
    g.ksteps = ks;
    g.ksteps_per_split = per;
    g.tiles_m = g.tiles_n = T;
    EpiSlab epi{t->cov_ws, (long long)n, nn, n, n, splits};
    const int wgs = upper_tiles(T) * splits;
    int side_wgs = kNumCU - wgs;
    side_wgs = side_wgs < 8 ? 8 : (side_wgs > 64 ? 64 : side_wgs);
    FastSide side{};
    side.n = n;
    side.mv[0] = SideMatvec{t->invsqrtC, nullptr, t->colsum, t->step, t->whitened, nullptr, t->rb, 0, side_wgs};
    e = launch<128, 128, 2, 4, 1, true, 3, EpiSlab, FastSide>(g, epi, upper_tiles(T), splits, 1, st, side, side_wgs);
    if (e) return e;
  }
  }

  // ---- covariance, paths, scaled planes ----
  const long long plane = (long long)n * t->ldn;
  const long long sm = plane * 2;
  uint16_t* planes = (uint16_t*)t->planes;
  uint16_t *Y0 = planes, *A2 = planes + sm, *Qp = planes + 2 * sm, *Pp = planes + 3 * sm;
  float* Tf = t->ns_f32 + 2 * nn;
  const float* inv_s = t->scal + 8;          // 1 / s, written by the covariance kernel
  const int T64 = (n + 63) / 64;
  {
    CovFast a{};
    a.ws = t->cov_ws;
    a.splits = splits;
    if (t->sharded) {
      a.packed = t->gram;
      a.poison = t->poison;
    }
    a.C = t->C; a.mean = t->mean; a.step = t->step; a.whitened = t->whitened; a.ps = t->ps; a.pc = t->pc;
    a.sigma = t->sigma; a.iter = t->iter_dev; a.prev_rec = t->prev_rec;
    a.k = PathConst{t->consts[0], t->consts[1], t->consts[2], t->consts[3], t->consts[4], t->consts[5], t->consts[6], t->consts[7]};
    a.C_o = t->C_o; a.mean_o = t->mean_o; a.ps_o = t->ps_o; a.pc_o = t->pc_o; a.scalars = t->scalars_o;
    a.y0 = Y0; a.ldp = t->ldn; a.plane = plane; a.scal = t->scal;
    a.n = n; a.tiles = T64;
    cov_fast_kernel<8, 1024><<<upper_tiles(T64), 1024, 0, st>>>(a);
    e = last_launch_status();
    if (e) return e;
  }

  // ---- the three products of the fifth-order step, with the side jobs ----
  const int tiles = upper_tiles(T64);
  int side_wgs = kNumCU - tiles;
  if (side_wgs < 24) side_wgs = 24;
  const int pw_wgs = 8, pr_wgs = 8, cond_wgs = 8;
  const bool draw = f->key != nullptr && f->z_next != nullptr && f->rows > 0;
  auto draw_side = [&](FastSide& s, int part, bool keys) {      // third `part` of the rows
    if (!draw) return;
    s.key = f->key;
    s.z = (uint16_t*)f->z_next;
    s.ldz = t->ldn;
    s.row_offset = f->row_offset;
    const int third = (f->rows + 2) / 3;
    s.row0 = part * third;
    s.row1 = part == 2 ? f->rows : (part + 1) * third;
    if (s.row1 > f->rows) s.row1 = f->rows;
    s.keys_out = keys ? f->keys_next : nullptr;
  };
  float* w = t->vec4 + n;               // C' v
  float* w2 = t->vec4 + 3 * n;          // C'^{-1/2} v2
  float* u = Tf;                        // C'^{-1/2} probe
  {
    // A2 = Y0 Y0 and q = (c2 I + c3 Y0) + c4 A2 from one product; Y0 = C' / s is read as the fp32 covariance
    auto epi = affine<2>(n, t->ldn);
    epi.nout = 2;
    epi.o[0].alpha = 1.f;
    for (int p = 0; p < 2; ++p) epi.o[0].pl[0][p] = A2 + p * plane;
    epi.o[1].alpha = kP4c4;
    epi.o[1].beta = kP4c3;
    epi.o[1].beta_dev = inv_s;
    epi.o[1].gamma = kP4c2;
    epi.o[1].cin[0] = t->C_o;
    for (int p = 0; p < 2; ++p) epi.o[1].pl[0][p] = Qp + p * plane;
    FastSide side{};
    side.n = n;
    side.mv[0] = SideMatvec{t->C_o, t->ctrl_vec, nullptr, nullptr, w, t->scal, 0, 1, pw_wgs};
    draw_side(side, 0, true);
    e = product<WM, WN, 2>(Y0, Y0, plane, t->ldn, n, epi, side, side_wgs, st);
    if (e) return e;
  }
  {
    // p = (c0 I + c1 Y0) + A2 q  (= C'^{-1/2} sqrt(s)): planes for the last product, fp32 scaled by 1/sqrt(s) as the output
    auto epi = affine<2>(n, t->ldn);
    epi.nout = 2;
    for (int o = 0; o < 2; ++o) {
      epi.o[o].alpha = 1.f;
      epi.o[o].beta = kP4c1;
      epi.o[o].beta_dev = inv_s;
      epi.o[o].gamma = kP4c0;
      epi.o[o].cin[0] = t->C_o;
    }
    for (int p = 0; p < 2; ++p) epi.o[0].pl[0][p] = Pp + p * plane;
    epi.o[1].f32[0] = t->inv_o;
    epi.o[1].scale[0] = t->scal + 11;
    FastSide side{};
    side.n = n;
    draw_side(side, 1, false);
    e = product<WM, WN, 2>(A2, Qp, plane, t->ldn, n, epi, side, side_wgs, st);
    if (e) return e;
  }
  {
    // sqrt C' = sqrt(s) Y0 p
    auto epi = affine<1>(n, t->ldn);
    epi.o[0].alpha = 1.f;
    epi.o[0].f32[0] = t->B_o;
    epi.o[0].scale[0] = t->scal + 12;
    epi.o[0].pl[0][0] = (uint16_t*)t->b16_o;
    FastSide side{};
    side.n = n;
    side.mv[0] = SideMatvec{t->inv_o, nullptr, nullptr, nullptr, u, nullptr, 0, 0, pr_wgs};
    if (t->ns_cond) side.mv[1] = SideMatvec{t->inv_o, t->ctrl_vec + n, nullptr, nullptr, w2, t->scal2, 0, 1, cond_wgs};
    draw_side(side, 2, false);
    e = product<WM, WN, 1>(Y0, Pp, plane, t->ldn, n, epi, side, side_wgs, st);
    if (e) return e;
  }
  int res_blocks = (n + 3) / 4;
  if (res_blocks > kPartRow) res_blocks = kPartRow;
  residual_kernel<<<res_blocks, 256, 0, st>>>(t->B_o, n, u, t->scal);
  Finish fin{};
  fin.scal = t->scal; fin.scal2 = t->scal2;
  fin.res_blocks = res_blocks; fin.pw_wgs = pw_wgs; fin.cond_wgs = cond_wgs; fin.cond = t->ns_cond; fin.n = n;
  fin.ldp = t->ldn;
  fin.prev_sqrt = t->B; fin.prev_inv = t->invsqrtC; fin.out_sqrt = t->B_o; fin.out_inv = t->inv_o;
  fin.out_b16 = (uint16_t*)t->b16_o;
  fin.w = w; fin.w2 = w2; fin.ctrl_vec = t->ctrl_vec; fin.prev_rec = t->prev_rec;
  fin.ctrl_vec_o = t->ctrl_vec_o; fin.rec_o = t->rec_o; fin.rec_host = t->rec_host;
  fin.count_in = f->count_in; fin.count_out = f->count_next;
  finish_kernel<<<kNumCU, 256, 0, st>>>(fin);
  return last_launch_status();
}

// The folded tell.  Preconditions (the caller falls back to evox_hip_cma_tell otherwise): a single-process generation, or
// the part of a row-sharded one that follows the all-reduce (summed mean shift in `step`, tile-packed summed rank-mu term);
// factorisation by the symmetric two-plane driver with ONE fifth-order step, estimate-scaled (not the safe norm bound),
// n a multiple of 4, and a previous record whose scale entry is valid (any earlier call of either driver).
EVOX_API int evox_hip_cma_tell_fast(const CmaTell* t, const CmaFast* f, void* stream) {
  if (t == nullptr || f == nullptr || !t->factor || t->ns_symmetric != 1 || t->ns_iters != 1 || t->ns_safe ||
      t->ns_planes != 2 || (t->n & 3) || t->n <= 0 || t->b16_o == nullptr || t->prev_rec == nullptr ||
      t->ctrl_vec == nullptr)
    return (int)hipErrorInvalidValue;
  if (t->ldn < (int64_t)((t->n + 63) / 64) * 64 || (t->ldn & 7) || ((uintptr_t)t->planes & 15) || ((uintptr_t)t->b16_o & 15) ||
      ((uintptr_t)t->ns_f32 & 15) || ((uintptr_t)t->C & 15) || ((uintptr_t)t->C_o & 15))
    return (int)hipErrorInvalidValue;
  if (t->sharded && (t->gram == nullptr || !t->gram_packed || ((uintptr_t)t->gram & 15) || t->step == nullptr))
    return (int)hipErrorInvalidValue;
  if (f->key != nullptr && (f->z_next == nullptr || f->keys_next == nullptr || ((uintptr_t)f->z_next & 7))) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  return f->wide ? tell_fast<4, 4>(t, f, st) : tell_fast<2, 4>(t, f, st);
}
