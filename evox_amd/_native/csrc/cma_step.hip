// One native call per half of a CMA-ES generation.
//
// A generation at lambda = 10^4, n = 10^3 is ~25 kernels of 3-35 us.  Issued one by one from the interpreter, each
// costs 5-10 us of host time (argument marshalling, allocator, stream lookup), which is more than the kernels take:
// the step was bound by the host.  The two entry points below issue the whole kernel sequence of `ask` and of `tell`
// from C++ on the caller's stream; the interpreter allocates the output tensors and keeps the policy decisions
// (which factorisation driver, how many iterations) that need values from earlier generations.
//
//   ask   (reference: es_variants/cma_es.py:128-135)  key split, Gaussian draws (unless drawn ahead), X = m + s Z S^T
//   tell  (reference: es_variants/cma_es.py:137-196)  rank sort, transposed factor of the selected rows + mean shift,
//         whitening matvec, path / step-size update, rank-mu covariance update, factorisation C^{1/2}, C^{-1/2}
#include "hip_common.h"
#include "cma_step.h"

extern "C" {
int evox_hip_split(const int32_t* keys, int32_t* out, int64_t nkeys, int64_t n, void* stream);
int evox_hip_split_count(const int32_t* key, int32_t* out, int64_t n, const float* count_in, float* count_out, void* stream);
int evox_hip_normal_bf16_pitched(const int32_t* key, void* out, int64_t rows, int64_t n, int64_t ld, int64_t row_offset,
                                 void* stream);
int evox_hip_bf16_sample(const void* Z, int64_t ldz, const void* S, int64_t lds, const float* mean, const float* sigma,
                         float* X, int64_t ldx, int M, int N, int K, void* stream);
int evox_hip_rank_sort_sliced(const float* keys, int n, int64_t* order, int64_t* rank_out, int* scratch, void* stream);
int evox_hip_cma_gather_t(const float* pop, int64_t ldpop, const int64_t* order, const float* mean, const float* rowscale,
                          const float* sigma, void* yt, int64_t ldy, float* colsum, int rb, float* step, int mu, int n,
                          int lam, void* stream);
int evox_hip_cma_paths(const float* mean, const float* step, const float* whitened, const float* ps, const float* pc,
                       const float* sigma, const float* iter, const float* consts, int n, float* mean_out, float* ps_out,
                       float* pc_out, float* scalars, void* stream);
int evox_hip_bf16_cov(const void* Yt, int64_t ldy, int n, int mu, const float* C, const float* pc, const float* alpha,
                      float c1, float cmu, float* out, float* ws, int64_t ws_floats, void* stream);
int evox_hip_cov_from_packed(const float* packed, int n, const float* C, const float* pc, const float* alpha, float c1,
                             float cmu, const float* poison, float* out, void* stream);
int evox_hip_ns_sqrt_planes(const float* C, int n, int64_t ldp, int nsp, int iters, int power_steps, int safe, int full,
                            void* planes, float* f32ws, float* vec, float* scal, const float* prev_sqrt, const float* prev_inv,
                            float* out_sqrt, float* out_inv, void* out_sqrt_bf16, void* stream);
int evox_hip_ns_sqrt(const float* C, int n, int iters, int power_steps, int safe, float* work, const float* eye, float* vec,
                     float* scal, const float* prev_sqrt, const float* prev_inv, float* out_sqrt, float* out_inv, int tune,
                     int first_step, void* out_sqrt_bf16, void* stream);
int evox_hip_ns_cond(const float* inv, int n, float* vec2, float* scal, float* scal2, void* stream);
}

namespace {

using namespace evox;

// y = A x for a row-major n x n matrix: one wavefront per row, 16-byte loads
// ctrl (optional): the 2 x n power vectors of the factorisation control, laid out into vec4 = [v | scratch | v2 | scratch2]
// on the way (a copy that would otherwise be a launch of its own).
__global__ __launch_bounds__(256) void matvec_kernel(const float* __restrict__ A, int n, const float* __restrict__ x,
                                                     float* __restrict__ y, const float* __restrict__ ctrl,
                                                     float* __restrict__ vec4) {
  if (ctrl != nullptr) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 2 * n; i += gridDim.x * blockDim.x)
      vec4[(i / n) * 2 * n + (i % n)] = ctrl[i];
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int row = blockIdx.x * 4 + wave; row < n; row += gridDim.x * 4) {
    const float* r = A + (size_t)row * n;
    float acc = 0.f;
    if ((n & 3) == 0) {
      const float4* r4 = reinterpret_cast<const float4*>(r);
      const float4* x4 = reinterpret_cast<const float4*>(x);
      for (int j = lane; j < (n >> 2); j += 64) {
        const float4 a = r4[j], v = x4[j];
        acc += (a.x * v.x + a.y * v.y) + (a.z * v.z + a.w * v.w);
      }
    } else {
      for (int j = lane; j < n; j += 64) acc += r[j] * x[j];
    }
    acc = wave_sum(acc);
    if (lane == 0) y[row] = acc;
  }
}

// small device-to-device moves of the control block (fill kernels, not memcpy nodes: the sequence may be recorded)
__global__ void copy_floats_kernel(const float* __restrict__ src, float* __restrict__ dst, int count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) dst[i] = src[i];
}
// The control vectors and the 7-float record of the call leave together.  Without a refresh of the condition
// estimates (`carry` set) the two estimates of the previous record are carried over.  rec_host (optional): the same
// record written straight into pinned host memory, where the iteration control reads it two calls later — no copy
// command, the write is visible once the kernel has completed.
__global__ void ctrl_out_kernel(const float* __restrict__ vec4, const float* __restrict__ scal, const float* __restrict__ carry,
                                float* __restrict__ ctrl, float* __restrict__ rec, float* __restrict__ rec_host, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 2 * n) ctrl[i] = vec4[(i / n) * 2 * n + (i % n)];
  if (i < 7) {
    const float v = (carry != nullptr && i < 2) ? carry[i] : scal[9 + i];
    rec[i] = v;
    if (rec_host != nullptr) rec_host[i] = v;
  }
}

// Row-sharded generation: the rank-mu term arrives summed over the ranks (gram = sum_r S_r^T S_r), so the covariance update
// is one elementwise pass, out = alpha C + c1 pc pc^T + cmu gram.  `poison` (optional, one float): a positive value —
// some rank had to drop selected rows — turns the result into NaNs on every rank alike.
__global__ __launch_bounds__(256) void cov_from_gram_kernel(const float* __restrict__ C, const float* __restrict__ gram,
                                                            const float* __restrict__ pc, const float* __restrict__ alpha,
                                                            float c1, float cmu, const float* __restrict__ poison,
                                                            float* __restrict__ out, int n) {
  const float a = *alpha;
  const bool bad = poison != nullptr && *poison > 0.f;
  const int64_t total = (int64_t)n * n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / n), c = (int)(i - (int64_t)r * n);
    const float v = fmaf(a, C[i], fmaf(c1, pc[r] * pc[c], cmu * gram[i]));
    out[i] = bad ? __builtin_nanf("") : v;
  }
}

// Order-preserving unsigned image of an fp32 key, the one the counting sort ranks by (-0 = +0, NaN after +inf).
__device__ __forceinline__ uint32_t sortable_key(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return 0xFFFFFFFEu;
  if (u == 0x80000000u) u = 0u;
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// Global ranks of this process's rows when every process has sorted its own fitness values before the all-gather.
// msg: world x 2 x local floats — row 2q the fitness of process q's rows, row 2q + 1 the same values sorted.  The global
// (stable: ties by global row index) rank of local row i is its rank among its own rows plus, for every other process,
// the number of that process's values in front of it: one binary search per process, all of them advanced together so
// that their loads are in flight at the same time.
constexpr int kSearchGroup = 8;
__global__ __launch_bounds__(256) void presorted_rank_kernel(const float* __restrict__ msg, int world, int me, int local,
                                                             const int64_t* __restrict__ own_rank, int* __restrict__ ranks_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= local) return;
  const uint32_t k = sortable_key(msg[(size_t)(2 * me) * local + i]);
  int total = (int)own_rank[i];
  int steps = 1;
  while ((1 << steps) <= local) ++steps;                 // a range of `local` entries is empty after this many halvings
  for (int q0 = 0; q0 < world; q0 += kSearchGroup) {
    int lo[kSearchGroup], hi[kSearchGroup];
#pragma unroll
    for (int u = 0; u < kSearchGroup; ++u) {
      const int q = q0 + u;
      lo[u] = 0;
      hi[u] = (q < world && q != me) ? local : 0;
    }
    for (int s = 0; s < steps; ++s) {
#pragma unroll
      for (int u = 0; u < kSearchGroup; ++u) {
        if (lo[u] < hi[u]) {
          const int q = q0 + u;
          const int mid = (lo[u] + hi[u]) >> 1;
          const uint32_t v = sortable_key(msg[(size_t)(2 * q + 1) * local + mid]);
          // values of a process in front of mine win ties (k <= 0xFFFFFFFE: k + 1 cannot wrap)
          const bool before = v < k + (uint32_t)(q < me);
          if (before) lo[u] = mid + 1; else hi[u] = mid;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < kSearchGroup; ++u) total += lo[u];
  }
  ranks_out[i] = total;
}

// The ranks come either from `ranking` (global rank of every individual) or, when `partial` is given, as the sum of
// `slices` partial counts per local row (evox_hip_rank_window_count).
__device__ __forceinline__ int local_rank(const int64_t* ranking, const int* partial, int slices, int lo, int local, int i) {
  if (!partial) return (int)ranking[lo + i];
  int r = 0;
  for (int s = 0; s < slices; ++s) r += partial[(size_t)s * local + i];
  return r;
}

// Row-sharded selection: of this rank's rows [lo, lo + local) those whose global rank is below mu are compacted, in row
// order, into `keep` slots (row index relative to lo, and the square root of the recombination weight of its rank); the
// unused slots get weight zero.  More than `keep` selected rows cannot be held: the count of such ranks is what
// `overflow[0]` carries into the all-reduced message (overflow[1..3] are cleared).
__global__ __launch_bounds__(1024) void select_local_kernel(const int64_t* __restrict__ ranking, const int* __restrict__ partial,
                                                            int slices, int lo, int local, int mu,
                                                            int keep, const float* __restrict__ sqrt_w_padded,
                                                            int64_t* __restrict__ order_out, float* __restrict__ w_out,
                                                            float* __restrict__ overflow) {
  // wavefront w owns the rows [w·per, (w+1)·per): 64 consecutive rows per step (coalesced), selected rows counted and
  // placed with ballots; one barrier for the offsets of the sixteen wavefronts.  Up to 16 steps per wavefront (16 384
  // rows) the ranks are loaded once, all loads in flight together, and kept in registers for the placement.
  __shared__ int s_tot[16];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int per = ((local + 15) / 16 + 63) / 64 * 64;
  const int first = min(w * per, local), last = min(first + per, local);
  constexpr int kHeld = 16;
  const bool held = per <= kHeld * 64;
  int rk[kHeld];
  int cnt = 0;
  if (held) {
#pragma unroll
    for (int u = 0; u < kHeld; ++u) {
      const int i = first + u * 64 + lane;
      rk[u] = i < last ? local_rank(ranking, partial, slices, lo, local, i) : mu;
    }
#pragma unroll
    for (int u = 0; u < kHeld; ++u) cnt += __popcll(__ballot(rk[u] < mu));
  } else {
    for (int i0 = first; i0 < last; i0 += 64) {
      const int i = i0 + lane;
      const bool sel = i < last && local_rank(ranking, partial, slices, lo, local, i) < mu;
      cnt += __popcll(__ballot(sel));
    }
  }
  if (lane == 0) s_tot[w] = cnt;
  __syncthreads();
  int pos = 0, total = 0;
  for (int q = 0; q < 16; ++q) {
    if (q < w) pos += s_tot[q];
    total += s_tot[q];
  }
  if (held) {
#pragma unroll
    for (int u = 0; u < kHeld; ++u) {
      const int i = first + u * 64 + lane;
      const int r = rk[u];
      const bool sel = r < mu;
      const unsigned long long m = __ballot(sel);
      const int at = pos + __popcll(m & ((1ull << lane) - 1ull));
      if (sel && at < keep) { order_out[at] = i; w_out[at] = sqrt_w_padded[r]; }
      pos += __popcll(m);
    }
  } else {
    for (int i0 = first; i0 < last; i0 += 64) {
      const int i = i0 + lane;
      const int r = i < last ? local_rank(ranking, partial, slices, lo, local, i) : mu;
      const bool sel = r < mu;
      const unsigned long long m = __ballot(sel);
      const int at = pos + __popcll(m & ((1ull << lane) - 1ull));
      if (sel && at < keep) { order_out[at] = i; w_out[at] = sqrt_w_padded[r]; }
      pos += __popcll(m);
    }
  }
  for (int k = total + (int)threadIdx.x; k < keep; k += 1024) { order_out[k] = 0; w_out[k] = 0.f; }
  if (threadIdx.x < 4) overflow[threadIdx.x] = (threadIdx.x == 0 && total > keep) ? 1.f : 0.f;   // the message's four trailing floats
}

}  // namespace

// ranking: global rank of every individual (int64, lam_total); sqrt_w_padded: lam_total floats (zero from mu on).
EVOX_API int evox_hip_cma_select_local(const int64_t* ranking, int lo, int local, int mu, int keep, const float* sqrt_w_padded,
                                       int64_t* order_out, float* w_out, float* overflow, void* stream) {
  if (local <= 0 || keep <= 0 || keep > local || lo < 0 || mu <= 0) return (int)hipErrorInvalidValue;
  select_local_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(ranking, nullptr, 0, lo, local, mu, keep, sqrt_w_padded, order_out,
                                                          w_out, overflow);
  return last_launch_status();
}

// The same from the fitness of the whole population (n_total keys): the local rows are ranked against all keys by
// counting (evox_hip_rank_window_count; scratch: evox_hip_rank_window_slices() x local ints), then compacted.
extern "C" int evox_hip_rank_window_count(const float* keys, int n, int row_lo, int rows_n, int* scratch, void* stream);

EVOX_API int evox_hip_cma_rank_select_local(const float* fitness, int n_total, int lo, int local, int mu, int keep,
                                            const float* sqrt_w_padded, int64_t* order_out, float* w_out, float* overflow,
                                            int* scratch, void* stream) {
  if (local <= 0 || keep <= 0 || keep > local || lo < 0 || mu <= 0 || lo + local > n_total) return (int)hipErrorInvalidValue;
  const int slices = evox_hip_rank_window_count(fitness, n_total, lo, local, scratch, stream);
  if (slices < 0) return -slices;
  select_local_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(nullptr, scratch, slices, 0, local, mu, keep, sqrt_w_padded,
                                                          order_out, w_out, overflow);
  return last_launch_status();
}

// The same from the pre-sorted fitness message (see presorted_rank_kernel); own_rank: rank of every local row among the
// local rows (int64), scratch: local ints.
EVOX_API int evox_hip_cma_presorted_select_local(const float* msg, int world, int me, int local, const int64_t* own_rank,
                                                 int mu, int keep, const float* sqrt_w_padded, int64_t* order_out,
                                                 float* w_out, float* overflow, int* scratch, void* stream) {
  if (local <= 0 || keep <= 0 || keep > local || world <= 0 || me < 0 || me >= world || mu <= 0)
    return (int)hipErrorInvalidValue;
  presorted_rank_kernel<<<(local + 255) / 256, 256, 0, (hipStream_t)stream>>>(msg, world, me, local, own_rank, scratch);
  select_local_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(nullptr, scratch, 1, 0, local, mu, keep, sqrt_w_padded, order_out,
                                                          w_out, overflow);
  return last_launch_status();
}

// y = A x (A: n x n row-major fp32): the whitening product of the path update
EVOX_API int evox_hip_matvec(const float* A, int n, const float* x, float* y, void* stream) {
  if (n <= 0) return (int)hipErrorInvalidValue;
  int mv_blocks = (n + 3) / 4;
  if (mv_blocks > kNumCU * 4) mv_blocks = kNumCU * 4;
  matvec_kernel<<<mv_blocks, 256, 0, (hipStream_t)stream>>>(A, n, x, y, nullptr, nullptr);
  return last_launch_status();
}

namespace {
int matvec_with_control(const float* A, int n, const float* x, float* y, const float* ctrl, float* vec4, hipStream_t st) {
  int mv_blocks = (n + 3) / 4;
  if (mv_blocks > kNumCU * 4) mv_blocks = kNumCU * 4;
  matvec_kernel<<<mv_blocks, 256, 0, st>>>(A, n, x, y, ctrl, vec4);
  return last_launch_status();
}
}  // namespace

struct CmaAsk {
  const int32_t* key;        // [2]
  int32_t* keys_out;         // [2][2]: next key, sampling key
  void* z;                   // rows x ldn bf16 draws
  const void* s16;           // n x ldn bf16 sampling factor
  const float* mean;
  const float* sigma;
  float* x;                  // rows x n fp32 out
  int64_t ldn, row_offset;
  int rows, n;
  int draw;                  // 0: z already holds this generation's draws
  int split;                 // 0: keys_out already holds the split of key
  // draws of the NEXT generation on a side stream (they depend on the key alone), all optional
  int wait_side;             // this generation's split and draws were made ahead: wait for them first
  int prefetch;              // make the next generation's now
  void* side;                // hipStream_t (default priority)
  void* ev_main;             // hipEvent_t pair created by evox_hip_event_create
  void* ev_side;
  int32_t* keys_next;        // [2][2]
  void* z_next;              // the draw buffer that is not `z`
  // generation counter on the device, advanced by the split kernel (optional)
  const float* count_in;
  float* count_out;
};

EVOX_API void* evox_hip_event_create() {
  hipEvent_t e = nullptr;
  if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
  return (void*)e;
}
EVOX_API void evox_hip_event_destroy(void* e) {
  if (e) (void)hipEventDestroy((hipEvent_t)e);
}

EVOX_API int evox_hip_cma_ask(const CmaAsk* a, void* stream) {
  int e = 0;
  hipStream_t main_st = (hipStream_t)stream;
  if (a->wait_side) {
    if (hipStreamWaitEvent(main_st, (hipEvent_t)a->ev_side, 0) != hipSuccess) return (int)hipErrorUnknown;
  }
  if (a->split) {
    e = evox_hip_split_count(a->key, a->keys_out, 2, a->count_in, a->count_out, stream);
    if (e) return e;
  } else if (a->count_out != nullptr) {
    e = evox_hip_split_count(a->key, a->keys_out, 0, a->count_in, a->count_out, stream);
    if (e) return e;
  }
  if (a->draw) {
    e = evox_hip_normal_bf16_pitched(a->keys_out + 2, a->z, a->rows, a->n, a->ldn, a->row_offset, stream);
    if (e) return e;
  }
  e = evox_hip_bf16_sample(a->z, a->ldn, a->s16, a->ldn, a->mean, a->sigma, a->x, a->n, a->rows, a->n, a->n, stream);
  if (e || !a->prefetch) return e;
  // The side stream starts after the product above has been issued: the other draw buffer was last read by the
  // previous generation's product (earlier on this stream), and the key it splits is in keys_out.
  hipStream_t side = (hipStream_t)a->side;
  if (hipEventRecord((hipEvent_t)a->ev_main, main_st) != hipSuccess) return (int)hipErrorUnknown;
  if (hipStreamWaitEvent(side, (hipEvent_t)a->ev_main, 0) != hipSuccess) return (int)hipErrorUnknown;
  e = evox_hip_split(a->keys_out, a->keys_next, 1, 2, a->side);
  if (e) return e;
  e = evox_hip_normal_bf16_pitched(a->keys_next + 2, a->z_next, a->rows, a->n, a->ldn, a->row_offset, a->side);
  if (e) return e;
  if (hipEventRecord((hipEvent_t)a->ev_side, side) != hipSuccess) return (int)hipErrorUnknown;
  return 0;
}

EVOX_API int evox_hip_cma_tell(const CmaTell* t, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int n = t->n;
  int e = 0;
  if (!t->sharded) {
    e = evox_hip_rank_sort_sliced(t->fitness, t->lam, t->order, nullptr, t->sort_scratch, stream);
    if (e) return e;
    e = evox_hip_cma_gather_t(t->pop, t->ldpop, t->order, t->mean, t->rowscale, t->sigma, t->yt, t->ldmu, t->colsum, t->rb,
                              t->step, t->mu, n, t->lam, stream);
    if (e) return e;
  } else if (t->gram == nullptr) {
    return (int)hipErrorInvalidValue;
  }
  e = matvec_with_control(t->invsqrtC, n, t->step, t->whitened, t->factor ? t->ctrl_vec : nullptr, t->vec4, st);
  if (e) return e;
  e = evox_hip_cma_paths(t->mean, t->step, t->whitened, t->ps, t->pc, t->sigma, t->iter_dev, t->consts, n, t->mean_o, t->ps_o,
                         t->pc_o, t->scalars_o, stream);
  if (e) return e;
  if (!t->sharded) {
    e = evox_hip_bf16_cov(t->yt, t->ldmu, n, t->mu, t->C, t->pc_o, t->scalars_o + 1, t->consts[6], t->consts[7], t->C_o,
                          t->cov_ws, t->cov_ws_floats, stream);
  } else if (t->gram_packed) {
    e = evox_hip_cov_from_packed(t->gram, n, t->C, t->pc_o, t->scalars_o + 1, t->consts[6], t->consts[7], t->poison, t->C_o,
                                 stream);
  } else {
    cov_from_gram_kernel<<<stream_grid((int64_t)n * n, 256), 256, 0, st>>>(t->C, t->gram, t->pc_o, t->scalars_o + 1,
                                                                           t->consts[6], t->consts[7], t->poison, t->C_o, n);
    e = last_launch_status();
  }
  if (e || !t->factor) return e;
  if (t->ns_symmetric) {
    // ns_symmetric: 1 mirrored plane products, 2 plain plane products in the iterations (ill-conditioned covariance)
    e = evox_hip_ns_sqrt_planes(t->C_o, n, t->ldn, t->ns_planes, t->ns_iters, t->ns_power, t->ns_safe,
                                t->ns_symmetric == 2, t->planes, t->ns_f32, t->vec4, t->scal, t->B, t->invsqrtC, t->B_o,
                                t->inv_o, t->b16_o, stream);
  } else {
    if (t->lib_work == nullptr || t->lib_eye == nullptr) return (int)hipErrorInvalidValue;
    e = evox_hip_ns_sqrt(t->C_o, n, t->ns_iters, t->ns_power, t->ns_safe, t->lib_work, t->lib_eye, t->vec4, t->scal, t->B,
                         t->invsqrtC, t->B_o, t->inv_o, t->ns_tune, t->ns_first_poly4, nullptr, stream);
  }
  if (e) return e;
  if (t->ns_cond) {
    e = evox_hip_ns_cond(t->inv_o, n, t->vec4 + 2 * n, t->scal, t->scal2, stream);
    if (e) return e;
  }
  ctrl_out_kernel<<<(2 * n + 255) / 256, 256, 0, st>>>(t->vec4, t->scal, t->ns_cond ? nullptr : t->prev_rec, t->ctrl_vec_o,
                                                      t->rec_o, t->rec_host, n);
  return last_launch_status();
}
