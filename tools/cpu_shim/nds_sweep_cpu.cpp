// The sweep non-dominated-sort kernels on the CPU under sanitizers, against a brute-force ranking.
//   g++ -std=c++20 -O1 -g -fsanitize=address,undefined -I tools/cpu_shim tools/cpu_shim/nds_sweep_cpu.cpp -o /tmp/nds_sweep_cpu -lpthread
#include "hip_common.h"
#include "../../evox_amd/_native/csrc/nds_sweep.hip"

#include <algorithm>
#include <array>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>

int main(int argc, char** argv) {
  const int n = argc > 1 ? std::atoi(argv[1]) : 3000;
  int npad = 2048, levels = 11;
  while (npad < n) { npad <<= 1; ++levels; }
  const int chunks = npad / 2048;
  std::mt19937 rng(1);
  std::uniform_int_distribution<int> d(0, 11);
  std::vector<std::array<float, 3>> pts(n);
  for (auto& p : pts) p = {d(rng) / 12.f, d(rng) / 12.f, d(rng) / 12.f};
  std::sort(pts.begin(), pts.end());
  std::vector<float> f2(npad, INFINITY), f3(npad, INFINITY);
  std::vector<int> alive(npad, 0), gsize(npad, 0), rep(n);
  for (int i = 0; i < n; ++i) {
    f2[i] = pts[i][1]; f3[i] = pts[i][2];
    if (argc > 2 && i == 7) f2[i] = NAN;          // what an unsanitised NaN objective does to the structure
    rep[i] = (i > 0 && pts[i] == pts[i - 1]) ? rep[i - 1] : i;
    gsize[rep[i]]++;
    alive[i] = rep[i] == i;
  }
  std::vector<int> ord((size_t)(levels + 1) * npad, -1), pos((size_t)levels * npad), rank(npad, -1), state(2 + n + 8, 0);
  std::vector<float> key((size_t)(levels + 1) * npad), pm((size_t)levels * npad), agg((size_t)levels * chunks), cpm((size_t)levels * chunks);
  shim::launch(dim3(npad / 256), dim3(256), [&] { sweep_level0_kernel(f2.data(), npad, ord.data(), key.data()); });
  for (int level = 0; level < levels; ++level)
    shim::launch(dim3(npad / 256), dim3(256), [&] {
      sweep_merge_kernel(ord.data() + (size_t)level * npad, key.data() + (size_t)level * npad, npad, level,
                         ord.data() + (size_t)(level + 1) * npad, key.data() + (size_t)(level + 1) * npad,
                         pos.data() + (size_t)level * npad);
    });
  // every level must hold a permutation of the indices: an unwritten slot would be read as an index by the scan pass
  for (int level = 0; level < levels; ++level) {
    std::vector<int> seen(npad, 0);
    for (int k = 0; k < npad; ++k) {
      const int e = ord[(size_t)level * npad + k];
      if (e < 0 || e >= npad || seen[e]++) { std::printf("level %d is not a permutation\n", level); return 2; }
    }
  }
  const int target = n;
  int rounds = 0;
  while (state[0] < target && rounds < n) {
    const int upper = levels - kChunkLog - 1;
    shim::launch(dim3(chunks, levels), dim3(256), [&] {
      sweep_scan_kernel(ord.data(), f3.data(), alive.data(), npad, pm.data(), agg.data(), state.data(), target);
    });
    shim::launch(dim3(upper > 0 ? upper : 1), dim3(256), [&] { sweep_chunks_kernel(agg.data(), cpm.data(), npad, levels, state.data(), target); });
    shim::launch(dim3(npad / 256), dim3(256), [&] {
      sweep_query_kernel(pos.data(), pm.data(), cpm.data(), f3.data(), alive.data(), gsize.data(), npad, levels, rounds, rank.data(),
                         state.data(), target);
    });
    ++rounds;
  }
  // brute force
  std::vector<int> want(n, -1);
  int left = n, r = 0;
  while (left > 0) {
    std::vector<int> front;
    for (int i = 0; i < n; ++i) {
      if (want[i] >= 0) continue;
      bool dom = false;
      for (int j = 0; j < n && !dom; ++j) {
        if (want[j] >= 0 || j == i) continue;
        const auto &a = pts[j], &b = pts[i];
        dom = a[0] <= b[0] && a[1] <= b[1] && a[2] <= b[2] && a != b;
      }
      if (!dom) front.push_back(i);
    }
    for (int i : front) want[i] = r;
    left -= (int)front.size();
    ++r;
  }
  int bad = 0;
  for (int i = 0; i < n; ++i) bad += rank[rep[i]] != want[i];
  std::printf("rounds %d (want %d)  ranked %d  mismatches %d\n", rounds, r, state[0], bad);
  return bad != 0;
}
