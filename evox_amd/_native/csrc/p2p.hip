// One-shot collectives over peer-mapped device memory (xGMI between GPUs of one node).
//
// The messages this framework exchanges every generation are small (fitness slices of a few KB,
// migrants) or mid-sized (the n + n² sufficient statistics of CMA-ES, 4 MB at n = 10³).  A ring
// serialises W−1 hops over single links; xGMI is point-to-point with a link to every peer, so the
// direct form is: every rank WRITES its contribution straight into every peer's receive slot (all
// links busy at once, one kernel launch), raises a flag there, and each rank then waits for the W
// flags of its own slot and copies (all-gather) or sums (all-reduce) what arrived.
//
// Large sums go in two shots instead (reduce-scatter, then all-gather, both in the direct form): rank r sends
// chunk p of its message to peer p only, sums the W pieces of its own chunk as they arrive, and writes the
// reduced chunk to every peer.  Each link then carries 2/W of the message instead of all of it.
//
// Each rank owns one buffer, exported through hipIpc and mapped by every peer:
//     flags[2 phases][2 slots][kMaxWorld] (sequence numbers) | data[2][W][max_bytes] | reduced[2][W][chunk_bytes]
// Two slots alternate.  Reuse is safe without an extra barrier: a rank starts collective s+1 only
// after it has seen every peer's flag for s, and a peer raises its flag for s only after its own
// stream has finished consuming slot (s−1)&1 — the slot that s+1 overwrites.
//
// The sequence number lives on the device: every kernel of a collective derives it from the same counter word, and the
// last workgroup of the collective's final kernel advances the counter.  No launch argument changes from one collective
// to the next, so a recorded HIP graph that contains collectives replays correctly (every process replays in step).
//
// Waiting is bounded: the wait kernel polls a limited number of times, then records a time-out in
// an error word (device copy and a host-visible copy), fills its share of the output with NaNs and finishes, so a
// missing peer can never leave waves spinning on the GPU nor hand uninitialised memory to the caller.  After a
// time-out the slot protocol is no longer in step: the group must not be used again.
#include <cstring>

#include "hip_common.h"

namespace {

constexpr int kMaxWorld = 64;
constexpr int kHeaderBytes = 4 * kMaxWorld * 4;          // flags of both phases and both slots
constexpr unsigned int kNaNBits = 0x7FC00000u;

// which = slot (messages, first phase) or 2 + slot (reduced chunks, second phase)
__device__ __forceinline__ unsigned int* flag_ptr(char* base, int which, int rank) {
  return reinterpret_cast<unsigned int*>(base) + which * kMaxWorld + rank;
}

// sequence number of the collective in flight (0 is the flags' initial value and never used) and its slot
__device__ __forceinline__ unsigned int next_seq(const unsigned int* seq_dev) {
  const unsigned int s = *seq_dev + 1u;
  return s == 0u ? 2u : s;                               // wrap-around keeps the slots alternating
}

// The last workgroup of a collective's final kernel: advance the device-side sequence number (stream order puts the next
// collective's first kernel behind this one).
__device__ __forceinline__ void advance_seq(unsigned int* seq_dev, unsigned int seq, unsigned int* done_c) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int finished = atomicAdd(done_c, 1u) + 1u;
    if (finished == gridDim.x) {
      *done_c = 0u;
      *seq_dev = seq;
    }
  }
}

// Threads 0..W-1 of a block poll the W flags (bounded); returns true on every thread when one of them gave up.
__device__ __forceinline__ bool wait_flags(char* base, int which, int world, unsigned int seq, long long max_polls,
                                           int* timed_out_lds) {
  if (threadIdx.x == 0) *timed_out_lds = 0;
  __syncthreads();
  if ((int)threadIdx.x < world) {
    unsigned int* f = flag_ptr(base, which, threadIdx.x);
    long long polls = 0;
    while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
      if (++polls >= max_polls) { *timed_out_lds = 1; break; }
      __builtin_amdgcn_s_sleep(32);
    }
  }
  __syncthreads();
  return *timed_out_lds != 0;
}

__device__ __forceinline__ void report_timeout(int* err, int* err_host) {
  if (threadIdx.x == 0) {
    atomicExch(err, 1);
    if (err_host != nullptr) __hip_atomic_store(err_host, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// grid (blocks, W): column p copies this rank's message (chunk_words == 0) or chunk p of it (words
// [p·chunk_words, min((p+1)·chunk_words, total))) into peer p's slot and, once every block of the column has
// finished, publishes the sequence number there.
__global__ __launch_bounds__(256) void p2p_put_kernel(const unsigned int* __restrict__ src, int64_t n_words,
                                                      char* const* __restrict__ peer_base, int rank,
                                                      int64_t slot_stride, int64_t max_bytes,
                                                      const unsigned int* __restrict__ seq_dev,
                                                      unsigned int* __restrict__ done, int64_t chunk_words) {
  const unsigned int seq = next_seq(seq_dev);
  const int slot = (int)(seq & 1u);
  const int p = blockIdx.y;
  char* base = peer_base[p];
  unsigned int* dst = reinterpret_cast<unsigned int*>(base + kHeaderBytes + slot * slot_stride + (int64_t)rank * max_bytes);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  if (chunk_words > 0) {
    const int64_t first = (int64_t)p * chunk_words;
    src += first;
    n_words = n_words - first < chunk_words ? n_words - first : chunk_words;     // may be <= 0 for the last ranks
    if (n_words < 0) n_words = 0;
  }
  if ((n_words & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
    const uint4* s4 = reinterpret_cast<const uint4*>(src);
    uint4* d4 = reinterpret_cast<uint4*>(dst);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n_words >> 2); i += stride) d4[i] = s4[i];
  } else {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += stride) dst[i] = src[i];
  }
  __threadfence_system();                               // the data is visible to the peer before the flag
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int finished = atomicAdd(&done[p], 1u) + 1u;
    if (finished == gridDim.x) {
      done[p] = 0u;                                     // ready for the next collective (stream-ordered)
      __threadfence_system();
      __hip_atomic_store(flag_ptr(base, slot, rank), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// Every block waits (bounded) for the W flags of this rank's slot, then takes part in the copy-out.
// reduce == 0: out[r·n_words + i] = slot[r][i] (all-gather); reduce == 1: out[i] = Σ_r slot[r][i] as fp32.
__global__ __launch_bounds__(256) void p2p_collect_kernel(char* __restrict__ base, int world,
                                                          int64_t slot_stride, int64_t max_bytes, int64_t n_words,
                                                          unsigned int* __restrict__ seq_dev, unsigned int* __restrict__ done_c,
                                                          unsigned int* __restrict__ out, int reduce,
                                                          long long max_polls, int* __restrict__ err,
                                                          int* __restrict__ err_host) {
  __shared__ int timed_out;
  const unsigned int seq = next_seq(seq_dev);
  const int slot = (int)(seq & 1u);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  if (wait_flags(base, slot, world, seq, max_polls, &timed_out)) {
    // this block's share of the output is poisoned, never left uninitialised; the host raises
    report_timeout(err, err_host);
    const int64_t total = reduce ? n_words : n_words * world;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) out[i] = kNaNBits;
    advance_seq(seq_dev, seq, done_c);
    return;
  }
  __threadfence_system();
  const char* data = base + kHeaderBytes + slot * slot_stride;
  if (reduce) {
    float* o = reinterpret_cast<float*>(out);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += stride) {
      float acc = 0.f;
      for (int r = 0; r < world; ++r)
        acc += __builtin_nontemporal_load(reinterpret_cast<const float*>(data + (int64_t)r * max_bytes) + i);
      o[i] = acc;
    }
  } else {
    for (int r = 0; r < world; ++r) {
      const unsigned int* s = reinterpret_cast<const unsigned int*>(data + (int64_t)r * max_bytes);
      for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += stride)
        out[(int64_t)r * n_words + i] = __builtin_nontemporal_load(s + i);
    }
  }
  advance_seq(seq_dev, seq, done_c);
}

// Second shot of the two-shot sum.  After the (bounded) wait for the W pieces of this rank's chunk, every block sums
// its share and writes it into the reduced area of every peer; the last block to finish publishes the sequence number
// to all of them.  A rank whose wait ran out sends NaNs — its peers must not wait for it in turn.
__global__ __launch_bounds__(256) void p2p_reduce_bcast_kernel(char* __restrict__ base, char* const* __restrict__ peer_base,
                                                               int world, int rank, int64_t slot_stride,
                                                               int64_t max_bytes, int64_t red_offset, int64_t chunk_stride,
                                                               int64_t len_words, const unsigned int* __restrict__ seq_dev,
                                                               long long max_polls,
                                                               unsigned int* __restrict__ done, int* __restrict__ err,
                                                               int* __restrict__ err_host) {
  __shared__ int timed_out;
  const unsigned int seq = next_seq(seq_dev);
  const int slot = (int)(seq & 1u);
  const bool bad = wait_flags(base, slot, world, seq, max_polls, &timed_out);
  if (bad) report_timeout(err, err_host);
  __threadfence_system();
  const char* data = base + kHeaderBytes + slot * slot_stride;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t red = red_offset + (int64_t)slot * world * chunk_stride + (int64_t)rank * chunk_stride;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < len_words; i += stride) {
    float acc = 0.f;
    for (int r = 0; r < world; ++r)
      acc += __builtin_nontemporal_load(reinterpret_cast<const float*>(data + (int64_t)r * max_bytes) + i);
    if (bad) acc = __uint_as_float(kNaNBits);
    for (int p = 0; p < world; ++p) reinterpret_cast<float*>(peer_base[p] + red)[i] = acc;
  }
  __threadfence_system();
  __syncthreads();
  __shared__ unsigned int finished;
  if (threadIdx.x == 0) finished = atomicAdd(done, 1u) + 1u;
  __syncthreads();
  if (finished == gridDim.x) {
    if (threadIdx.x == 0) *done = 0u;
    __threadfence_system();
    if ((int)threadIdx.x < world)
      __hip_atomic_store(flag_ptr(peer_base[threadIdx.x], 2 + slot, rank), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// Third step: wait for the W reduced chunks and lay them out: out[r·chunk_words + i] = reduced[r][i].
__global__ __launch_bounds__(256) void p2p_collect_chunks_kernel(char* __restrict__ base, int world,
                                                                 int64_t red_offset, int64_t chunk_stride,
                                                                 int64_t chunk_words, int64_t n_words,
                                                                 unsigned int* __restrict__ seq_dev,
                                                                 unsigned int* __restrict__ done_c,
                                                                 unsigned int* __restrict__ out, long long max_polls,
                                                                 int* __restrict__ err, int* __restrict__ err_host) {
  __shared__ int timed_out;
  const unsigned int seq = next_seq(seq_dev);
  const int slot = (int)(seq & 1u);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  if (wait_flags(base, 2 + slot, world, seq, max_polls, &timed_out)) {
    report_timeout(err, err_host);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += stride) out[i] = kNaNBits;
    advance_seq(seq_dev, seq, done_c);
    return;
  }
  __threadfence_system();
  const char* red = base + red_offset + (int64_t)slot * world * chunk_stride;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += stride) {
    const int64_t r = i / chunk_words, k = i - r * chunk_words;
    out[i] = __builtin_nontemporal_load(reinterpret_cast<const unsigned int*>(red + r * chunk_stride) + k);
  }
  advance_seq(seq_dev, seq, done_c);
}

}  // namespace

EVOX_API int evox_hip_p2p_header_bytes() { return kHeaderBytes; }
EVOX_API int evox_hip_p2p_max_world() { return kMaxWorld; }

// Receive buffers are uncached device memory when the runtime offers it (a peer's writes must not be
// hidden by this GPU's L2); plain device memory otherwise.  Zero-filled: flags start at sequence 0.
EVOX_API int evox_hip_p2p_alloc(void** ptr, int64_t bytes) {
  hipError_t e = hipExtMallocWithFlags(ptr, (size_t)bytes, hipDeviceMallocUncached);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    e = hipMalloc(ptr, (size_t)bytes);
    if (e != hipSuccess) return (int)e;
  }
  e = hipMemset(*ptr, 0, (size_t)bytes);
  if (e != hipSuccess) return (int)e;
  return (int)hipDeviceSynchronize();
}

EVOX_API int evox_hip_p2p_free(void* ptr) { return (int)hipFree(ptr); }

EVOX_API int evox_hip_p2p_export(void* ptr, void* handle_out, int handle_bytes) {
  if (handle_bytes < (int)sizeof(hipIpcMemHandle_t)) return (int)hipErrorInvalidValue;
  hipIpcMemHandle_t h;
  hipError_t e = hipIpcGetMemHandle(&h, ptr);
  if (e != hipSuccess) return (int)e;
  memcpy(handle_out, &h, sizeof(h));
  return 0;
}

EVOX_API int evox_hip_p2p_handle_bytes() { return (int)sizeof(hipIpcMemHandle_t); }

EVOX_API int evox_hip_p2p_import(const void* handle, void** ptr) {
  hipIpcMemHandle_t h;
  memcpy(&h, handle, sizeof(h));
  return (int)hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess);
}

EVOX_API int evox_hip_p2p_close(void* ptr) { return (int)hipIpcCloseMemHandle(ptr); }

// peer_base: device array of W buffer addresses as seen from this process (own buffer at [rank])
// seq_dev: the group's sequence counter on the device (one zero-initialised word; see the head of this file)
EVOX_API int evox_hip_p2p_put(const void* src, int64_t n_words, void* peer_base, int world, int rank,
                              int64_t slot_stride, int64_t max_bytes, const void* seq_dev, void* done, int64_t chunk_words,
                              void* stream) {
  if (n_words <= 0 || world < 1 || world > kMaxWorld || chunk_words < 0) return (int)hipErrorInvalidValue;
  const int64_t per_peer = chunk_words > 0 ? chunk_words : n_words;
  if (per_peer * 4 > max_bytes || (chunk_words > 0 && (chunk_words & 3))) return (int)hipErrorInvalidValue;
  int blocks = (int)((per_peer / 4 + 255) / 256);
  if (blocks < 1) blocks = 1;
  if (blocks > 64) blocks = 64;                         // W columns share the chip
  p2p_put_kernel<<<dim3(blocks, world), 256, 0, (hipStream_t)stream>>>(
      (const unsigned int*)src, n_words, (char* const*)peer_base, rank, slot_stride, max_bytes,
      (const unsigned int*)seq_dev, (unsigned int*)done, chunk_words);
  return evox::last_launch_status();
}

// done_c: one zeroed counter word of the group (last-workgroup detection of the collective's final kernel)
EVOX_API int evox_hip_p2p_collect(void* own_base, int world, int64_t slot_stride, int64_t max_bytes,
                                  int64_t n_words, void* seq_dev, void* done_c, void* out, int reduce, long long max_polls,
                                  void* err, void* err_host, void* stream) {
  if (n_words <= 0 || world < 1 || world > kMaxWorld || max_polls < 1) return (int)hipErrorInvalidValue;
  const int64_t work = reduce ? n_words : n_words * world;
  int blocks = (int)((work + 1023) / 1024);
  if (blocks < 1) blocks = 1;
  if (blocks > 512) blocks = 512;
  p2p_collect_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>((char*)own_base, world, slot_stride, max_bytes, n_words,
                                                             (unsigned int*)seq_dev, (unsigned int*)done_c,
                                                             (unsigned int*)out, reduce, max_polls, (int*)err,
                                                             (int*)err_host);
  return evox::last_launch_status();
}

// Second and third step of the two-shot sum (the first is evox_hip_p2p_put with chunk_words > 0).
// red_offset: byte offset of the reduced area in every rank's buffer; done2: one zeroed counter.
EVOX_API int evox_hip_p2p_two_shot_finish(void* own_base, void* peer_base, int world, int rank, int64_t slot_stride,
                                          int64_t max_bytes, int64_t red_offset, int64_t chunk_stride, int64_t chunk_words,
                                          int64_t n_words, void* seq_dev, void* done_c, void* out, long long max_polls,
                                          void* done2, void* err, void* err_host, void* stream) {
  if (n_words <= 0 || world < 1 || world > kMaxWorld || max_polls < 1 || chunk_words <= 0 || (chunk_words & 3) ||
      chunk_words * 4 > chunk_stride || chunk_words * 4 > max_bytes || chunk_words * world < n_words || rank < 0 || rank >= world)
    return (int)hipErrorInvalidValue;
  int64_t len = n_words - (int64_t)rank * chunk_words;
  len = len < 0 ? 0 : (len > chunk_words ? chunk_words : len);
  int blocks = (int)((len + 1023) / 1024);
  blocks = blocks < 1 ? 1 : (blocks > 256 ? 256 : blocks);
  p2p_reduce_bcast_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>((char*)own_base, (char* const*)peer_base, world, rank,
                                                                  slot_stride, max_bytes, red_offset, chunk_stride, len,
                                                                  (const unsigned int*)seq_dev, max_polls,
                                                                  (unsigned int*)done2, (int*)err, (int*)err_host);
  int e = evox::last_launch_status();
  if (e) return e;
  blocks = (int)((n_words + 1023) / 1024);
  blocks = blocks < 1 ? 1 : (blocks > 512 ? 512 : blocks);
  p2p_collect_chunks_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>((char*)own_base, world, red_offset, chunk_stride,
                                                                    chunk_words, n_words, (unsigned int*)seq_dev,
                                                                    (unsigned int*)done_c, (unsigned int*)out, max_polls,
                                                                    (int*)err, (int*)err_host);
  return evox::last_launch_status();
}

// device address of a pinned host allocation (the host-visible copy of the error word)
EVOX_API int evox_hip_host_device_ptr(void* host, void** dev) { return (int)hipHostGetDevicePointer(dev, host, 0); }
