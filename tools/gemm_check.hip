// Stand-alone correctness and timing check of the bf16 NT GEMM family (csrc/bf16_gemm.h) on one GPU.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Xclang -target-feature -Xclang -packed-fp32-ops \
//         -I evox_amd/_native/csrc tools/gemm_check.hip -o gpurun_out/gemm_check && gpurun_out/gemm_check
//
// Every configuration is compared against a one-thread-per-element fp64 reference kernel on random,
// asymmetric data (a transposed or mis-mapped fragment cannot pass), then timed with events.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "bf16_gemm.h"

using namespace evox;
using namespace evox::bg;

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      printf("HIP error %d (%s) at %s:%d\n", (int)e_, hipGetErrorString(e_), __FILE__, __LINE__); \
      fflush(stdout);                                                              \
      exit(2);                                                                     \
    }                                                                              \
  } while (0)

static uint16_t h_bf16(float v) {
  uint32_t u;
  memcpy(&u, &v, 4);
  return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
static float h_f32(uint16_t b) {
  uint32_t u = (uint32_t)b << 16;
  float v;
  memcpy(&v, &u, 4);
  return v;
}

// ref[m][n] = sum_k (sum_p A_p[m][k]) (sum_p B_p[n][k]) in fp64
__global__ void ref_kernel(const uint16_t* const* a, const uint16_t* const* b, int np, long long lda, long long ldb, int M,
                           int N, int K, double* out) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long long)M * N) return;
  const int m = (int)(e / N), n = (int)(e % N);
  double acc = 0;
  for (int k = 0; k < K; ++k) {
    double x = 0, y = 0;
    for (int p = 0; p < np; ++p) {
      x += (double)bf16_bits_to_f32(a[p][m * lda + k]);
      y += (double)bf16_bits_to_f32(b[p][n * ldb + k]);
    }
    acc += x * y;
  }
  out[e] = acc;
}

static int g_fail = 0;
static int g_pitch_or = 1;   // 1: odd number of 128-byte units per row (rows spread over the L2 channels)

struct Mat {   // fp32 master [rows][cols] + np bf16 planes with pitch ld (zero pad)
  int rows, cols;
  long long ld;
  std::vector<float> h;
  std::vector<uint16_t*> d;     // device planes
  uint16_t** dlist = nullptr;   // device array of plane pointers
};

static Mat make_mat(int rows, int cols, int np, std::mt19937& rng, float scale, bool symmetric = false) {
  Mat m;
  m.rows = rows;
  m.cols = cols;
  m.ld = (long long)(((cols + 63) / 64) | g_pitch_or) * 64;
  m.h.resize((size_t)rows * cols);
  std::normal_distribution<float> nd(0.f, scale);
  for (auto& v : m.h) v = nd(rng);
  if (symmetric)
    for (int i = 0; i < rows; ++i)
      for (int j = 0; j < i; ++j) m.h[(size_t)i * cols + j] = m.h[(size_t)j * cols + i];
  std::vector<std::vector<uint16_t>> planes(np, std::vector<uint16_t>((size_t)rows * m.ld, 0));
  for (int i = 0; i < rows; ++i)
    for (int j = 0; j < cols; ++j) {
      float rem = m.h[(size_t)i * cols + j];
      for (int p = 0; p < np; ++p) {
        const uint16_t b = h_bf16(rem);
        planes[p][(size_t)i * m.ld + j] = b;
        rem -= h_f32(b);
      }
    }
  for (int p = 0; p < np; ++p) {
    uint16_t* dp;
    CK(hipMalloc(&dp, planes[p].size() * 2));
    CK(hipMemcpy(dp, planes[p].data(), planes[p].size() * 2, hipMemcpyHostToDevice));
    m.d.push_back(dp);
  }
  CK(hipMalloc(&m.dlist, sizeof(uint16_t*) * np));
  CK(hipMemcpy(m.dlist, m.d.data(), sizeof(uint16_t*) * np, hipMemcpyHostToDevice));
  return m;
}

static std::vector<double> reference(const Mat& A, const Mat& B, int np, int K) {
  double* d;
  const size_t total = (size_t)A.rows * B.rows;
  CK(hipMalloc(&d, total * 8));
  ref_kernel<<<(unsigned)((total + 255) / 256), 256>>>(A.dlist, B.dlist, np, A.ld, B.ld, A.rows, B.rows, K, d);
  CK(hipDeviceSynchronize());
  std::vector<double> h(total);
  CK(hipMemcpy(h.data(), d, total * 8, hipMemcpyDeviceToHost));
  CK(hipFree(d));
  return h;
}

template <class F>
static float time_us(F&& f, int reps = 20) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) f();
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1000.f / reps;
}



// ---- sampling shape ----------------------------------------------------------------------------
static int g_probe = 0, g_zero_pitch = 0;

template <int BM, int BN, int WM, int WN, int ST = 2>
static void check_sample(const Mat& Z, const Mat& S, const std::vector<double>& ref, int K) {
  const int M = Z.rows, N = S.rows;
  std::vector<float> mean(N);
  for (int i = 0; i < N; ++i) mean[i] = 0.01f * i - 3.f;
  const float sigma = 0.7f;
  float *dmean, *dsigma, *dX;
  CK(hipMalloc(&dmean, N * 4));
  CK(hipMalloc(&dsigma, 4));
  CK(hipMalloc(&dX, (size_t)M * N * 4));
  CK(hipMemcpy(dmean, mean.data(), N * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dsigma, &sigma, 4, hipMemcpyHostToDevice));
  CK(hipMemset(dX, 0xff, (size_t)M * N * 4));
  Args<1> g{};
  g.a[0][0] = Z.d[0];
  g.b[0][0] = S.d[0];
  g.lda = (g_zero_pitch & 1) ? 0 : Z.ld;
  g.ldb = (g_zero_pitch & 2) ? 0 : S.ld;
  g.probe = g_probe;
  g.M = M;
  g.N = N;
  g.ksteps = (K + 63) / 64;
  g.ksteps_per_split = g.ksteps;
  g.tiles_m = (M + BM - 1) / BM;
  g.tiles_n = (N + BN - 1) / BN;
  EpiSample epi{dX, dmean, dsigma, (long long)N, M, N};
  auto run = [&]() {
    int rc = launch<BM, BN, WM, WN, 1, false, ST, EpiSample>(g, epi, g.tiles_m * g.tiles_n, 1, 1, 0);
    if (rc) { printf("launch failed %d\n", rc); exit(3); }
  };
  printf("sample  st%d ld%lld tile %3dx%3d waves %dx%d  M=%d N=%d K=%d tiles=%d ...", ST, Z.ld, BM, BN, WM, WN, M, N, K, g.tiles_m * g.tiles_n);
  fflush(stdout);
  run();
  CK(hipDeviceSynchronize());
  std::vector<float> X((size_t)M * N);
  CK(hipMemcpy(X.data(), dX, X.size() * 4, hipMemcpyDeviceToHost));
  double worst = 0;
  for (size_t e = 0; e < X.size(); ++e) {
    const double want = mean[e % N] + sigma * ref[e];
    const double err = std::fabs(X[e] - want);
    if (!(err <= worst)) worst = err;
  }
  const float us = time_us(run);
  const bool ok = worst < 2e-3 || g_probe || g_zero_pitch;
  if (g_probe || g_zero_pitch) printf(" [probe %d zero-pitch %d]", g_probe, g_zero_pitch);
  printf(" max|err| %.3g  %s  %.1f us  %.0f TFLOP/s\n", worst, ok ? "ok" : "FAIL", us, 2.0 * M * N * K / us * 1e-6);
  fflush(stdout);
  if (!ok) ++g_fail;
  CK(hipFree(dmean));
  CK(hipFree(dsigma));
  CK(hipFree(dX));
}

// ---- symmetric rank-k update, upper tiles into K-split slabs --------------------------------------
template <int BT, int WM, int WN, int ST = 2>
static void check_syrk(const Mat& Y, const std::vector<double>& ref, int K, int splits_req) {
  const int n = Y.rows;
  const int T = (n + BT - 1) / BT;
  const int ksteps = (K + 63) / 64;
  const int per = (ksteps + splits_req - 1) / splits_req;
  const int splits = (ksteps + per - 1) / per;
  float* ws;
  const size_t slab = (size_t)n * n;
  CK(hipMalloc(&ws, slab * splits * 4));
  CK(hipMemset(ws, 0, slab * splits * 4));
  Args<1> g{};
  g.a[0][0] = Y.d[0];
  g.b[0][0] = Y.d[0];
  g.lda = g.ldb = Y.ld;
  g.M = g.N = n;
  g.ksteps = ksteps;
  g.ksteps_per_split = per;
  g.tiles_m = g.tiles_n = T;
  EpiSlab epi{ws, (long long)n, (long long)slab, n, n, splits};
  auto run = [&]() {
    int rc = launch<BT, BT, WM, WN, 1, true, ST, EpiSlab>(g, epi, upper_tiles(T), splits, 1, 0);
    if (rc) { printf("launch failed %d\n", rc); exit(3); }
  };
  printf("syrk    st%d tile %3dx%3d waves %dx%d  n=%d K=%d splits=%d wgs=%d ...", ST, BT, BT, WM, WN, n, K, splits, upper_tiles(T) * splits);
  fflush(stdout);
  run();
  CK(hipDeviceSynchronize());
  std::vector<float> h(slab * splits);
  CK(hipMemcpy(h.data(), ws, h.size() * 4, hipMemcpyDeviceToHost));
  double worst = 0;
  for (int i = 0; i < n; ++i)
    for (int j = i; j < n; ++j) {
      // only tiles with tile_i <= tile_j are written; inside them every element is
      double s = 0;
      for (int q = 0; q < splits; ++q) s += h[q * slab + (size_t)i * n + j];
      const double err = std::fabs(s - ref[(size_t)i * n + j]);
      if (!(err <= worst)) worst = err;
    }
  const float us = time_us(run);
  const bool ok = worst < 2e-2;
  printf(" max|err| %.3g  %s  %.1f us  %.0f TFLOP/s (half)\n", worst, ok ? "ok" : "FAIL", us, 1.0 * n * n * K / us * 1e-6);
  fflush(stdout);
  if (!ok) ++g_fail;
  CK(hipFree(ws));
}

// ---- symmetric products from split planes with the affine epilogue -------------------------------
template <int NP, int BT, int WM, int WN, int ST = 2>
static void check_symm(const Mat& A, const Mat& B, const std::vector<double>& ref, int K, int batch) {
  const int n = A.rows;
  const int T = (n + BT - 1) / BT;
  const long long ldp = A.ld;
  float* outf[2];
  uint16_t* outp[2][3];
  float* cin;
  std::vector<float> hc((size_t)n * n);
  for (size_t e = 0; e < hc.size(); ++e) hc[e] = 0.001f * (float)(e % 977);
  CK(hipMalloc(&cin, hc.size() * 4));
  CK(hipMemcpy(cin, hc.data(), hc.size() * 4, hipMemcpyHostToDevice));
  for (int z = 0; z < 2; ++z) {
    CK(hipMalloc(&outf[z], (size_t)n * n * 4));
    CK(hipMemset(outf[z], 0xff, (size_t)n * n * 4));
    for (int p = 0; p < 3; ++p) {
      CK(hipMalloc(&outp[z][p], (size_t)n * ldp * 2));
      CK(hipMemset(outp[z][p], 0, (size_t)n * ldp * 2));
    }
  }
  Args<NP> g{};
  for (int z = 0; z < 2; ++z)
    for (int p = 0; p < NP; ++p) {
      g.a[z][p] = A.d[p];
      g.b[z][p] = B.d[p];
    }
  g.lda = A.ld;
  g.ldb = B.ld;
  g.probe = g_probe;
  g.M = g.N = n;
  g.ksteps = (K + 63) / 64;
  g.ksteps_per_split = g.ksteps;
  g.tiles_m = g.tiles_n = T;
  EpiAffine<NP> epi{};
  epi.nout = 1;
  epi.o[0].alpha = -0.5f;
  epi.o[0].beta = 2.0f;
  epi.o[0].gamma = 1.5f;
  for (int z = 0; z < 2; ++z) {
    epi.o[0].f32[z] = outf[z];
    epi.o[0].cin[z] = cin;
    for (int p = 0; p < NP; ++p) epi.o[0].pl[z][p] = outp[z][p];
  }
  epi.ldf = n;
  epi.ldp = ldp;
  epi.M = epi.N = n;
  epi.keep = nullptr;
  auto run = [&]() {
    int rc = launch<BT, BT, WM, WN, NP, true, ST, EpiAffine<NP>>(g, epi, upper_tiles(T), 1, batch, 0);
    if (rc) { printf("launch failed %d\n", rc); exit(3); }
  };
  printf("symm    st%d planes %d tile %3dx%3d waves %dx%d  n=%d K=%d batch=%d wgs=%d ...", ST, NP, BT, BT, WM, WN, n, K, batch,
         upper_tiles(T) * batch);
  fflush(stdout);
  run();
  CK(hipDeviceSynchronize());
  double worst = 0, worst_pl = 0, asym = 0;
  for (int z = 0; z < batch; ++z) {
    std::vector<float> h((size_t)n * n);
    CK(hipMemcpy(h.data(), outf[z], h.size() * 4, hipMemcpyDeviceToHost));
    std::vector<std::vector<uint16_t>> pl(NP, std::vector<uint16_t>((size_t)n * ldp));
    for (int p = 0; p < NP; ++p) CK(hipMemcpy(pl[p].data(), outp[z][p], pl[p].size() * 2, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) {
        // everything below the diagonal mirrors the element above it
        const bool below = i > j;
        const int r = below ? j : i, c = below ? i : j;
        const double want = -0.5 * ref[(size_t)r * n + c] + 2.0 * hc[(size_t)r * n + c] + (i == j ? 1.5 : 0.0);
        const double got = h[(size_t)i * n + j];
        const double err = std::fabs(got - want);
        if (!(err <= worst)) worst = err;
        double sum = 0;
        for (int p = 0; p < NP; ++p) sum += h_f32(pl[p][(size_t)i * ldp + j]);
        const double e2 = std::fabs(sum - got);
        if (!(e2 <= worst_pl)) worst_pl = e2;
        {
          const double e3 = std::fabs((double)h[(size_t)i * n + j] - (double)h[(size_t)j * n + i]);
          if (!(e3 <= asym)) asym = e3;
        }
      }
  }
  const float us = time_us(run);
  const double tol = 2e-5;
  const bool ok = (worst < tol && asym == 0.0) || g_probe;
  if (g_probe) printf(" [probe %d]", g_probe);
  printf(" max|err| %.3g planes-vs-f32 %.3g asym %.3g  %s  %.1f us\n", worst, worst_pl, asym, ok ? "ok" : "FAIL", us);
  fflush(stdout);
  if (!ok) ++g_fail;
  CK(hipFree(cin));
  for (int z = 0; z < 2; ++z) {
    CK(hipFree(outf[z]));
    for (int p = 0; p < 3; ++p) CK(hipFree(outp[z][p]));
  }
}

int main(int argc, char** argv) {
  const bool quick = argc > 1 && atoi(argv[1]) == 1;
  // timing experiments (results are not checked): GEMM_PROBE = 1 stage the first steps only, 2 skip the multiply,
  // 4 skip the epilogue (sums of these)
  if (const char* e = getenv("GEMM_PROBE")) g_probe = atoi(e);
  std::mt19937 rng(1234);
  {
    // small odd shapes first: edge handling
    Mat Z = make_mat(300, 200, 1, rng, 1.f), S = make_mat(200, 200, 1, rng, 0.05f);
    auto ref = reference(Z, S, 1, 200);
    check_sample<160, 256, 2, 4>(Z, S, ref, 200);
    check_sample<160, 256, 2, 4, 3>(Z, S, ref, 200);
    check_sample<128, 128, 2, 2, 4>(Z, S, ref, 200);
    check_sample<64, 64, 2, 2, 5>(Z, S, ref, 200);
  }
  if (!quick) {
    const int lam = 10000, n = 1000;
    Mat Z = make_mat(lam, n, 1, rng, 1.f), S = make_mat(n, n, 1, rng, 0.05f);
    auto ref = reference(Z, S, 1, n);
    check_sample<160, 256, 2, 4>(Z, S, ref, n);
    check_sample<160, 256, 2, 4, 3>(Z, S, ref, n);
  }
  {
    const int n = quick ? 200 : 1000, mu = quick ? 700 : 5000;
    Mat Y = make_mat(n, mu, 1, rng, 0.02f);
    auto ref = reference(Y, Y, 1, mu);
    check_syrk<128, 2, 4, 3>(Y, ref, mu, 7);
    check_syrk<128, 2, 4, 4>(Y, ref, mu, 7);
    check_syrk<64, 2, 2, 3>(Y, ref, mu, 4);
  }
  {
    const int n = quick ? 200 : 1000;
    Mat A = make_mat(n, n, 3, rng, 0.05f, true), B = make_mat(n, n, 3, rng, 0.05f, true);
    auto ref3 = reference(A, B, 3, n);
    check_symm<3, 64, 2, 4, 2>(A, B, ref3, n, 1);
    check_symm<3, 64, 2, 4, 3>(A, B, ref3, n, 1);
    auto ref2 = reference(A, B, 2, n);
    check_symm<2, 64, 2, 4, 2>(A, B, ref2, n, 1);
    check_symm<2, 64, 2, 4, 3>(A, B, ref2, n, 1);
    check_symm<2, 64, 2, 4, 4>(A, B, ref2, n, 1);
    check_symm<2, 64, 2, 4, 5>(A, B, ref2, n, 1);
    check_symm<2, 64, 2, 4, 4>(A, B, ref2, n, 2);
    check_symm<2, 64, 2, 2, 4>(A, B, ref2, n, 1);
    check_symm<2, 64, 4, 4, 4>(A, B, ref2, n, 1);
    auto ref1 = reference(A, B, 1, n);
    check_symm<1, 64, 2, 4, 5>(A, B, ref1, n, 2);
  }
  printf(g_fail ? "FAILED: %d configuration(s)\n" : "all configurations ok\n", g_fail);
  return g_fail ? 1 : 0;
}
