// Planar articulated-body physics, one environment per lane: the sub-steps of one environment step in one launch.
//
// The model (problems/neuroevolution/reinforcement_learning/planar_physics.py) is a tree of L <= 7 rigid rods in the
// x-z plane with generalized coordinates q = (x, z, phi0, theta_1 .. theta_{L-1}).  Every point of a link is
// base + sum_k W[p][k] u(alpha_k) with u the unit vectors of the links' absolute angles and W a static weight row, so the
// Jacobian of a point is two small contractions with the ancestor matrix.  A lane assembles the (L+2) x (L+2) mass matrix
// and the generalized forces (velocity products, gravity, joint springs / dampers / limits, actuators, penalty ground
// contact with smoothed Coulomb friction, resistive drag) of ITS environment in registers, solves by an unrolled
// Cholesky factorisation and integrates semi-implicitly; locked base coordinates become identity rows.  Nothing is
// shared between lanes: 4096 environments are 64 independent wavefronts.
//
// Reference role: the environment step the reference obtains from Brax (problems/neuroevolution/reinforcement_learning/
// brax.py:45-96); the formulas are those of PlanarEngine.acceleration / .step, which is the oracle of the tests.
#include "hip_common.h"

using namespace evox;

namespace {

constexpr int kLM = 7;       // links
constexpr int kCM = 8;       // contact points
constexpr int kAM = 8;       // actuators
// the dampers of the penalty forces fade in with the penetration (forces continuous in the state)
constexpr float kContactRamp = 1.0e-3f;    // metres
constexpr float kLimitRamp = 1.0e-2f;      // radians

// offsets into the parameter block (floats; counts are stored as floats)
// task block: [0] recipe, [1] control range, then per recipe
//   1 locomotion: [2] forward weight, [3] control cost, [4] healthy bonus, [5] terminates, [6] z low, [7] z high,
//                 [8] angle low, [9] angle high, [10] leading coordinates left out of the observation, [11] velocity
//                 clip, [12] seconds per environment step
//   2 cart with one pole: [2] largest pole angle
//   3 cart with two poles, 4 two-link reacher (target in aux): [16 .. 22] weight row of the tip
constexpr int kOffTask = 440, kTaskFloats = 24, kOffTip = kOffTask + 16;
constexpr int kOffHead = 0, kOffFree = 16, kOffAnc = 32, kOffWcom = 96, kOffWcon = 160, kOffMass = 224, kOffLen = 232,
              kOffRest = 240, kOffMconst = 256, kOffLo = 352, kOffHi = 368, kOffDamp = 384, kOffStiff = 400,
              kOffActDof = 416, kOffActGear = 424, kParamFloats = 464;

template <int L>
__global__ __launch_bounds__(64) void planar_step_kernel(const float* __restrict__ P, float* __restrict__ q_io,
                                                         float* __restrict__ qd_io, const float* __restrict__ ctrl,
                                                         int batch, float* __restrict__ obs_out,
                                                         float* __restrict__ total, uint8_t* __restrict__ done_acc,
                                                         const float* __restrict__ aux) {
  constexpr int N = L + 2;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  const int ncon = (int)P[kOffHead + 1], nact = (int)P[kOffHead + 2], substeps = (int)P[kOffHead + 3];
  const float dt = P[kOffHead + 4], vmax = P[kOffHead + 5], grav = P[kOffHead + 6], kc = P[kOffHead + 7],
              cc = P[kOffHead + 8], mu = P[kOffHead + 9], klim = P[kOffHead + 10], clim = P[kOffHead + 11],
              drag_t = P[kOffHead + 12], drag_n = P[kOffHead + 13], drag_a = P[kOffHead + 14];
  const bool has_drag = drag_t != 0.f || drag_n != 0.f || drag_a != 0.f;
  const bool task = obs_out != nullptr;                  // the locomotion recipe: observation, reward, termination
  const float crange = task ? P[kOffTask + 1] : 3.0e38f;
  float x[N], v[N], qact[N], x_before[N];
#pragma unroll
  for (int d = 0; d < N; ++d) {
    x[d] = q_io[(long long)b * N + d];
    v[d] = qd_io[(long long)b * N + d];
    x_before[d] = x[d];
    qact[d] = 0.f;
  }
  float ctrl_sq = 0.f;
  for (int a = 0; a < nact; ++a) {
    const int dof = (int)P[kOffActDof + a];
    const float u = fminf(fmaxf(ctrl[(long long)b * nact + a], -crange), crange);
    ctrl_sq = fmaf(u, u, ctrl_sq);
    const float f = P[kOffActGear + a] * u;
#pragma unroll
    for (int d = 0; d < N; ++d) qact[d] += d == dof ? f : 0.f;
  }
  for (int s = 0; s < substeps; ++s) {
    float ux[L], uz[L], ad[L];
#pragma unroll
    for (int i = 0; i < L; ++i) {
      float al = P[kOffRest + i], w = 0.f;
#pragma unroll
      for (int k = 0; k < L; ++k) {
        const float a = P[kOffAnc + i * kLM + k];
        al = fmaf(a, x[2 + k], al);
        w = fmaf(a, v[2 + k], w);
      }
      sincosf(al, &uz[i], &ux[i]);
      ad[i] = w;
    }
    // M (lower triangle, row-major full storage) and Q
    float M[N][N], Q[N];
#pragma unroll
    for (int d = 0; d < N; ++d) {
#pragma unroll
      for (int e = 0; e < N; ++e) M[d][e] = e <= d ? P[kOffMconst + d * (kLM + 2) + e] : 0.f;
      const float lo = P[kOffLo + d], hi = P[kOffHi + d];
      const float below = fmaxf(lo - x[d], 0.f), above = fmaxf(x[d] - hi, 0.f);
      const float outside = fminf((below + above) * (1.f / kLimitRamp), 1.f);     // the damper fades in
      Q[d] = qact[d] - P[kOffDamp + d] * v[d] - P[kOffStiff + d] * x[d] + klim * (below - above) - clim * outside * v[d];
    }
    // Jacobian of the point with weight row W: dP/dtheta_j = sum_k W[k] anc[k][j] uperp_k
    auto jacobian = [&](const float* W, float* Jx, float* Jz) {
      float gx[L], gz[L];
#pragma unroll
      for (int k = 0; k < L; ++k) {
        gx[k] = -W[k] * uz[k];
        gz[k] = W[k] * ux[k];
      }
#pragma unroll
      for (int j = 0; j < L; ++j) {
        float sx = 0.f, sz = 0.f;
#pragma unroll
        for (int k = 0; k < L; ++k) {
          const float a = P[kOffAnc + k * kLM + j];
          sx = fmaf(a, gx[k], sx);
          sz = fmaf(a, gz[k], sz);
        }
        Jx[j] = sx;
        Jz[j] = sz;
      }
    };
#pragma unroll
    for (int p = 0; p < L; ++p) {
      const float* W = P + kOffWcom + p * kLM;
      float Jx[L], Jz[L];
      jacobian(W, Jx, Jz);
      float a0x = 0.f, a0z = 0.f;
#pragma unroll
      for (int k = 0; k < L; ++k) {
        const float w2 = W[k] * ad[k] * ad[k];
        a0x = fmaf(-w2, ux[k], a0x);
        a0z = fmaf(-w2, uz[k], a0z);
      }
      const float m = P[kOffMass + p];
      M[0][0] += m;
      M[1][1] += m;
      const float bx = a0x, bz = a0z + grav;
      Q[0] -= m * bx;
      Q[1] -= m * bz;
#pragma unroll
      for (int j = 0; j < L; ++j) {
        M[2 + j][0] = fmaf(m, Jx[j], M[2 + j][0]);
        M[2 + j][1] = fmaf(m, Jz[j], M[2 + j][1]);
#pragma unroll
        for (int i = 0; i <= j; ++i) M[2 + j][2 + i] = fmaf(m, fmaf(Jx[j], Jx[i], Jz[j] * Jz[i]), M[2 + j][2 + i]);
        Q[2 + j] -= m * fmaf(Jx[j], bx, Jz[j] * bz);
      }
      if (has_drag) {
        float vcx = v[0], vcz = v[1];
#pragma unroll
        for (int j = 0; j < L; ++j) {
          vcx = fmaf(Jx[j], v[2 + j], vcx);
          vcz = fmaf(Jz[j], v[2 + j], vcz);
        }
        const float len = P[kOffLen + p];
        const float vt = vcx * ux[p] + vcz * uz[p], vn = -vcx * uz[p] + vcz * ux[p];
        const float fx = -len * (drag_t * vt * ux[p] - drag_n * vn * uz[p]);
        const float fz = -len * (drag_t * vt * uz[p] + drag_n * vn * ux[p]);
        const float tau = -drag_a * ad[p] * len * len * len * (1.f / 12.f);
        Q[0] += fx;
        Q[1] += fz;
#pragma unroll
        for (int j = 0; j < L; ++j) Q[2 + j] += fmaf(Jx[j], fx, Jz[j] * fz) + P[kOffAnc + p * kLM + j] * tau;
      }
    }
    for (int c = 0; c < ncon; ++c) {
      const float* W = P + kOffWcon + c * kLM;
      float pz = x[1];
#pragma unroll
      for (int k = 0; k < L; ++k) pz = fmaf(W[k], uz[k], pz);
      if (pz < 0.f) {
        float Jx[L], Jz[L];
        jacobian(W, Jx, Jz);
        float vx = v[0], vz = v[1];
#pragma unroll
        for (int j = 0; j < L; ++j) {
          vx = fmaf(Jx[j], v[2 + j], vx);
          vz = fmaf(Jz[j], v[2 + j], vz);
        }
        const float fn = fmaxf(kc * (-pz) - cc * vz * fminf(-pz * (1.f / kContactRamp), 1.f), 0.f);
        const float ft = -mu * fn * tanhf(vx * 20.f);
        Q[0] += ft;
        Q[1] += fn;
#pragma unroll
        for (int j = 0; j < L; ++j) Q[2 + j] += fmaf(Jx[j], ft, Jz[j] * fn);
      }
    }
    // locked base coordinates: identity rows
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      if (P[kOffFree + d] == 0.f) {
#pragma unroll
        for (int e = 0; e < N; ++e) {
          if (e < d) M[d][e] = 0.f;
          if (e > d) M[e][d] = 0.f;
        }
        M[d][d] = 1.f;
        Q[d] = 0.f;
      }
    }
    // Cholesky M = G G^T in place (lower), then the two triangular solves
#pragma unroll
    for (int j = 0; j < N; ++j) {
      float d = M[j][j];
#pragma unroll
      for (int t = 0; t < j; ++t) d = fmaf(-M[j][t], M[j][t], d);
      const float inv = rsqrtf(d);
      M[j][j] = d * inv;
#pragma unroll
      for (int i = j + 1; i < N; ++i) {
        float sacc = M[i][j];
#pragma unroll
        for (int t = 0; t < j; ++t) sacc = fmaf(-M[i][t], M[j][t], sacc);
        M[i][j] = sacc * inv;
      }
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
      float sacc = Q[i];
#pragma unroll
      for (int t = 0; t < i; ++t) sacc = fmaf(-M[i][t], Q[t], sacc);
      Q[i] = sacc / M[i][i];
    }
#pragma unroll
    for (int i = N - 1; i >= 0; --i) {
      float sacc = Q[i];
#pragma unroll
      for (int t = i + 1; t < N; ++t) sacc = fmaf(-M[t][i], Q[t], sacc);
      Q[i] = sacc / M[i][i];
    }
#pragma unroll
    for (int d = 0; d < N; ++d) {
      v[d] = fminf(fmaxf(fmaf(dt, Q[d], v[d]), -vmax), vmax);
      x[d] = fmaf(dt, v[d], x[d]);
    }
  }
  if (task) {
    // a state that left the numbers ends the episode where it was, without reward
    bool finite = true;
#pragma unroll
    for (int d = 0; d < N; ++d) finite = finite && fabsf(x[d]) < 3.0e38f && fabsf(v[d]) < 3.0e38f;   // false for NaN too
    if (!finite) {
#pragma unroll
      for (int d = 0; d < N; ++d) {
        x[d] = x_before[d];
        v[d] = 0.f;
      }
    }
    const float* T = P + kOffTask;
    const int recipe = (int)T[0];
    float reward;
    bool done_now;
    if (recipe == 1) {
      const bool terminates = T[5] != 0.f;
      const bool healthy = x[1] > T[6] && x[1] < T[7] && x[2] > T[8] && x[2] < T[9];
      const float vx = (x[0] - x_before[0]) / T[12];
      reward = T[2] * vx + ((healthy || !terminates) ? T[4] : 0.f) - T[3] * ctrl_sq;
      done_now = terminates && !healthy;
      const int skip = (int)T[10];
      const float clip = T[11];
      float* o = obs_out + (long long)b * (2 * N - skip);
      int w = 0;
#pragma unroll
      for (int d = 0; d < N; ++d)
        if (d >= skip) o[w++] = x[d];
#pragma unroll
      for (int d = 0; d < N; ++d) o[w++] = fminf(fmaxf(v[d], -clip), clip);
    } else if (recipe == 2) {
      // cart with one pole (L = 2): observation (x, theta, xdot, thetadot), one point per step while the pole stands
      const bool up = fabsf(x[N - 1]) <= T[2];
      reward = up ? 1.f : 0.f;
      done_now = !up;
      float* o = obs_out + (long long)b * 4;
      o[0] = x[0]; o[1] = x[N - 1]; o[2] = v[0]; o[3] = v[N - 1];
    } else {
      // the tip of the chain after the step
      float tx = x[0], tz = x[1];
#pragma unroll
      for (int i = 0; i < L; ++i) {
        float al = P[kOffRest + i];
#pragma unroll
        for (int k = 0; k < L; ++k) al = fmaf(P[kOffAnc + i * kLM + k], x[2 + k], al);
        float sn, cs;
        sincosf(al, &sn, &cs);
        tx = fmaf(P[kOffTip + i], cs, tx);
        tz = fmaf(P[kOffTip + i], sn, tz);
      }
      if (recipe == 3) {
        // cart with two poles (L = 3): (x, sin, sin, cos, cos, clipped velocities); 10 minus distance and velocity penalties
        const float t1 = x[N > 3 ? 3 : 0], t2 = x[N - 1];
        const float dist = 0.01f * tx * tx + (tz - 1.2f) * (tz - 1.2f);
        const float v1 = v[N > 3 ? 3 : 0];
        const float vel = 1e-3f * v1 * v1 + 5e-3f * v[N - 1] * v[N - 1];
        reward = 10.f - dist - vel;
        done_now = tz <= 1.f;
        float* o = obs_out + (long long)b * 8;
        o[0] = x[0]; o[1] = sinf(t1); o[2] = sinf(t2); o[3] = cosf(t1); o[4] = cosf(t2);
        o[5] = fminf(fmaxf(v[0], -10.f), 10.f); o[6] = fminf(fmaxf(v1, -10.f), 10.f); o[7] = fminf(fmaxf(v[N - 1], -10.f), 10.f);
      } else {
        // two-link reacher (L = 2): (cos, cos, sin, sin, target, velocities, tip - target, 0); minus distance minus controls squared
        const float gx = aux[(long long)b * 2], gz = aux[(long long)b * 2 + 1];
        const float dx = tx - gx, dz = tz - gz;
        reward = -sqrtf(dx * dx + dz * dz) - ctrl_sq;
        done_now = false;
        float* o = obs_out + (long long)b * 11;
        o[0] = cosf(x[2]); o[1] = cosf(x[N - 1]); o[2] = sinf(x[2]); o[3] = sinf(x[N - 1]);
        o[4] = gx; o[5] = gz; o[6] = v[2]; o[7] = v[N - 1]; o[8] = dx; o[9] = dz; o[10] = 0.f;
      }
    }
    if (!finite) reward = 0.f;
    done_now = done_now || !finite;
    if (!done_acc[b]) total[b] += reward;
    if (done_now) done_acc[b] = 1;
  }
#pragma unroll
  for (int d = 0; d < N; ++d) {
    q_io[(long long)b * N + d] = x[d];
    qd_io[(long long)b * N + d] = v[d];
  }
}

template <int L>
int launch_planar(const float* P, float* q, float* qd, const float* ctrl, int batch, float* obs, float* total,
                  uint8_t* done, const float* aux, hipStream_t st) {
  planar_step_kernel<L><<<(batch + 63) / 64, 64, 0, st>>>(P, q, qd, ctrl, batch, obs, total, done, aux);
  return last_launch_status();
}

}  // namespace

EVOX_API int evox_hip_planar_param_floats() { return kParamFloats; }

static int planar_dispatch(const float* params, int links, int contacts, int actuators, float* q, float* qd,
                           const float* ctrl, int batch, float* obs, float* total, uint8_t* done, const float* aux,
                           void* stream) {
  if (links < 1 || links > kLM || contacts < 0 || contacts > kCM || actuators < 0 || actuators > kAM || batch <= 0 ||
      params == nullptr || q == nullptr || qd == nullptr || (actuators > 0 && ctrl == nullptr))
    return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  switch (links) {
    case 1: return launch_planar<1>(params, q, qd, ctrl, batch, obs, total, done, aux, st);
    case 2: return launch_planar<2>(params, q, qd, ctrl, batch, obs, total, done, aux, st);
    case 3: return launch_planar<3>(params, q, qd, ctrl, batch, obs, total, done, aux, st);
    case 4: return launch_planar<4>(params, q, qd, ctrl, batch, obs, total, done, aux, st);
    case 5: return launch_planar<5>(params, q, qd, ctrl, batch, obs, total, done, aux, st);
    case 6: return launch_planar<6>(params, q, qd, ctrl, batch, obs, total, done, aux, st);
    default: return launch_planar<7>(params, q, qd, ctrl, batch, obs, total, done, aux, st);
  }
}

// One environment step (the model's sub-steps) of `batch` environments in place.  params: the block laid out as the
// offsets above (ops/planar.py builds it), on the device; q, qd: batch x (links + 2); ctrl: batch x actuators.
EVOX_API int evox_hip_planar_step(const float* params, int links, int contacts, int actuators, float* q, float* qd,
                                  const float* ctrl, int batch, void* stream) {
  return planar_dispatch(params, links, contacts, actuators, q, qd, ctrl, batch, nullptr, nullptr, nullptr, nullptr, stream);
}

// The same with the recipe of the task block applied in the kernel: controls clipped to the task's range, the observation
// written to obs (batch x the task's observation size), the step's reward added to total[b] unless done[b] was set before
// the step, done[b] set when the episode ends (the task's rule, or the state left the numbers: then it is put back and
// frozen).  aux: per-environment data of the recipe (the reacher's targets, batch x 2) or null.  recipe / links must
// agree (2: two links, 3: three links, 4: two links and aux).
EVOX_API int evox_hip_planar_step_task(const float* params, int links, int contacts, int actuators, float* q, float* qd,
                                       const float* ctrl, int batch, float* obs, float* total, uint8_t* done,
                                       const float* aux, int recipe, void* stream) {
  if (obs == nullptr || total == nullptr || done == nullptr || recipe < 1 || recipe > 4) return (int)hipErrorInvalidValue;
  if ((recipe == 2 && links != 2) || (recipe == 3 && links != 3) || (recipe == 4 && (links != 2 || aux == nullptr)))
    return (int)hipErrorInvalidValue;
  return planar_dispatch(params, links, contacts, actuators, q, qd, ctrl, batch, obs, total, done, aux, stream);
}
