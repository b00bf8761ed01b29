// Spatial articulated-body physics, one environment per lane: the sub-steps of one environment step in one launch.
//
// The model (problems/neuroevolution/reinforcement_learning/spatial_physics.py) is a free-floating root body with a tree
// of NB - 1 <= 8 bodies on hinge joints: q = (p, unit quaternion, theta_1 ..), qdot = (v, omega (world), thetadot_1 ..).
// A lane walks the tree of ITS environment once per sub-step (world rotation, frame origin, hinge axis, angular velocity
// and the accelerations at qddot = 0 of every body; these per-body quantities live in LDS, one slot per lane, because the
// parent of a body is a run-time index), then adds every body's share of the mass matrix and of the generalized forces
// from its Jacobian columns — I3, -[r - p]x and a_k x (r - j_k) for the hinges between root and body — in registers,
// adds joint springs / dampers / limits / actuators and the penalty ground contact, solves the (6 + J)-order system by
// an unrolled Cholesky factorisation and integrates semi-implicitly (the quaternion by its derivative, re-normalised).
//
// Reference role: the environment step the reference obtains from Brax for `ant`
// (problems/neuroevolution/reinforcement_learning/brax.py:45-96); the formulas are those of SpatialEngine.acceleration /
// .step_tensor, the oracle of the tests.
#include "hip_common.h"

using namespace evox;

namespace {

constexpr int kBM = 9;        // bodies
constexpr int kCM = 12;       // contact points
constexpr int kAM = 8;        // actuators
constexpr float kContactRamp = 1.0e-3f;
constexpr float kLimitRamp = 1.0e-2f;

// parameter block (floats; indices and counts stored as floats)
//   head [0..15]: bodies, contacts, actuators, substeps, dt, vmax, gravity, kc, cc, mu, klim, clim
//   body i at kOffBody + 16 i: parent, mass, com(3), inertia(3), joint_pos(3), axis(3)
//   anc[i][k] at kOffAnc + 9 i + k
//   joint j at kOffJoint + 8 j: lo, hi, damping, stiffness, armature
//   actuator a at kOffAct + 2 a: joint, gear
//   contact c at kOffCon + 4 c: body, point(3)
//   task at kOffTask: recipe (1 = ant), control range, forward weight, control cost, healthy bonus, z low, z high,
//                     velocity clip, seconds per environment step
constexpr int kOffBody = 16, kOffAnc = kOffBody + 16 * kBM, kOffJoint = kOffAnc + kBM * kBM, kOffAct = kOffJoint + 8 * (kBM - 1),
              kOffCon = kOffAct + 2 * kAM, kOffTask = kOffCon + 4 * kCM, kParamFloats = kOffTask + 16;

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float comp(V3 a, int c) { return c == 0 ? a.x : (c == 1 ? a.y : a.z); }

// per-lane storage in LDS: slot s of this lane
#define SLOT(s) lds[(s) * 64 + lane]
// slots of body i: R (9, row-major), x (3), a (3), w (3), alpha0 (3), ao0 (3)
constexpr int kBodySlots = 24;

template <int NB>
__global__ __launch_bounds__(64) void spatial_step_kernel(const float* __restrict__ P, float* __restrict__ pos_io,
                                                          float* __restrict__ quat_io, float* __restrict__ theta_io,
                                                          float* __restrict__ vel_io, const float* __restrict__ ctrl,
                                                          int batch, float* __restrict__ obs_out,
                                                          float* __restrict__ total, uint8_t* __restrict__ done_acc) {
  constexpr int J = NB - 1, N = 6 + J;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x;
  const int b = blockIdx.x * 64 + lane;
  const bool active = b < batch;
  const int bb = active ? b : batch - 1;                  // idle lanes shadow the last environment and store nothing
  const int ncon = (int)P[1], nact = (int)P[2], substeps = (int)P[3];
  const float dt = P[4], vmax = P[5], grav = P[6], kc = P[7], cc = P[8], mu = P[9], klim = P[10], clim = P[11];
  const bool task = obs_out != nullptr;
  const float crange = task ? P[kOffTask + 1] : 3.0e38f;

  float p[3], q[4], th[J > 0 ? J : 1], v[N];
#pragma unroll
  for (int c = 0; c < 3; ++c) p[c] = pos_io[(long long)bb * 3 + c];
#pragma unroll
  for (int c = 0; c < 4; ++c) q[c] = quat_io[(long long)bb * 4 + c];
#pragma unroll
  for (int j = 0; j < J; ++j) th[j] = theta_io[(long long)bb * J + j];
#pragma unroll
  for (int d = 0; d < N; ++d) v[d] = vel_io[(long long)bb * N + d];
  const float x_before = p[0];
  float tau_act[J > 0 ? J : 1];
#pragma unroll
  for (int j = 0; j < J; ++j) tau_act[j] = 0.f;
  float ctrl_sq = 0.f;
  for (int a = 0; a < nact; ++a) {
    const int joint = (int)P[kOffAct + 2 * a];
    const float u = fminf(fmaxf(ctrl[(long long)bb * nact + a], -crange), crange);
    ctrl_sq = fmaf(u, u, ctrl_sq);
    const float f = P[kOffAct + 2 * a + 1] * u;
#pragma unroll
    for (int j = 0; j < J; ++j) tau_act[j] += j == joint ? f : 0.f;
  }

  for (int s = 0; s < substeps; ++s) {
    // ---- the tree: rotation, origin, axis, angular velocity, accelerations at qddot = 0 ----
    {
      const float w = q[0], x = q[1], y = q[2], z = q[3];
      const float R0[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                           2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                           2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)};
#pragma unroll
      for (int e = 0; e < 9; ++e) SLOT(e) = R0[e];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        SLOT(9 + c) = p[c];
        SLOT(12 + c) = 0.f;
        SLOT(15 + c) = v[3 + c];
        SLOT(18 + c) = 0.f;
        SLOT(21 + c) = 0.f;
      }
    }
#pragma unroll
    for (int i = 1; i < NB; ++i) {
      const float* B = P + kOffBody + 16 * i;
      const int par = (int)B[0] * kBodySlots, me = i * kBodySlots;
      float Rp[9];
#pragma unroll
      for (int e = 0; e < 9; ++e) Rp[e] = SLOT(par + e);
      const V3 xp = {SLOT(par + 9), SLOT(par + 10), SLOT(par + 11)};
      const V3 wp = {SLOT(par + 15), SLOT(par + 16), SLOT(par + 17)};
      const V3 alp = {SLOT(par + 18), SLOT(par + 19), SLOT(par + 20)};
      const V3 aop = {SLOT(par + 21), SLOT(par + 22), SLOT(par + 23)};
      const V3 ax = {B[11], B[12], B[13]}, jp = {B[8], B[9], B[10]};
      const V3 a = {Rp[0] * ax.x + Rp[1] * ax.y + Rp[2] * ax.z, Rp[3] * ax.x + Rp[4] * ax.y + Rp[5] * ax.z,
                    Rp[6] * ax.x + Rp[7] * ax.y + Rp[8] * ax.z};
      const V3 r = {Rp[0] * jp.x + Rp[1] * jp.y + Rp[2] * jp.z, Rp[3] * jp.x + Rp[4] * jp.y + Rp[5] * jp.z,
                    Rp[6] * jp.x + Rp[7] * jp.y + Rp[8] * jp.z};
      // Rodrigues: rotation by theta about the axis (parent frame), R = Rp (I + s K + (1 - c) K^2)
      float sn, cs;
      sincosf(th[i - 1], &sn, &cs);
      const float oc = 1.f - cs;
      const float Kx = ax.x, Ky = ax.y, Kz = ax.z;
      const float A[9] = {cs + oc * Kx * Kx, oc * Kx * Ky - sn * Kz, oc * Kx * Kz + sn * Ky,
                          oc * Ky * Kx + sn * Kz, cs + oc * Ky * Ky, oc * Ky * Kz - sn * Kx,
                          oc * Kz * Kx - sn * Ky, oc * Kz * Ky + sn * Kx, cs + oc * Kz * Kz};
#pragma unroll
      for (int rr = 0; rr < 3; ++rr)
#pragma unroll
        for (int c2 = 0; c2 < 3; ++c2)
          SLOT(me + 3 * rr + c2) = Rp[3 * rr] * A[c2] + Rp[3 * rr + 1] * A[3 + c2] + Rp[3 * rr + 2] * A[6 + c2];
      const V3 xi = xp + r;
      const V3 rate = v[6 + i - 1] * a;
      const V3 wi = wp + rate;
      const V3 ali = alp + cross(wp, rate);
      const V3 aoi = aop + cross(alp, r) + cross(wp, cross(wp, r));
      SLOT(me + 9) = xi.x; SLOT(me + 10) = xi.y; SLOT(me + 11) = xi.z;
      SLOT(me + 12) = a.x; SLOT(me + 13) = a.y; SLOT(me + 14) = a.z;
      SLOT(me + 15) = wi.x; SLOT(me + 16) = wi.y; SLOT(me + 17) = wi.z;
      SLOT(me + 18) = ali.x; SLOT(me + 19) = ali.y; SLOT(me + 20) = ali.z;
      SLOT(me + 21) = aoi.x; SLOT(me + 22) = aoi.y; SLOT(me + 23) = aoi.z;
    }
    // ---- mass matrix (lower triangle) and generalized forces ----
    float M[N][N], Q[N];
#pragma unroll
    for (int d = 0; d < N; ++d) {
#pragma unroll
      for (int e = 0; e < N; ++e) M[d][e] = 0.f;
      Q[d] = 0.f;
    }
    const V3 base = {p[0], p[1], p[2]};
    // linear Jacobian of the world point r on body `body` (columns as three N-vectors), for static or run-time bodies
    auto jacobian = [&](V3 r, const float* anc_row, float (&Jx)[N], float (&Jy)[N], float (&Jz)[N]) {
      const V3 rb = r - base;
      Jx[0] = 1.f; Jx[1] = 0.f; Jx[2] = 0.f;
      Jy[0] = 0.f; Jy[1] = 1.f; Jy[2] = 0.f;
      Jz[0] = 0.f; Jz[1] = 0.f; Jz[2] = 1.f;
      // -[rb]x : velocity omega x rb
      Jx[3] = 0.f;   Jx[4] = rb.z;  Jx[5] = -rb.y;
      Jy[3] = -rb.z; Jy[4] = 0.f;   Jy[5] = rb.x;
      Jz[3] = rb.y;  Jz[4] = -rb.x; Jz[5] = 0.f;
#pragma unroll
      for (int k = 1; k < NB; ++k) {
        const float on = anc_row[k];
        const V3 ak = {SLOT(k * kBodySlots + 12), SLOT(k * kBodySlots + 13), SLOT(k * kBodySlots + 14)};
        const V3 xk = {SLOT(k * kBodySlots + 9), SLOT(k * kBodySlots + 10), SLOT(k * kBodySlots + 11)};
        const V3 col = on * cross(ak, r - xk);
        Jx[5 + k] = col.x; Jy[5 + k] = col.y; Jz[5 + k] = col.z;
      }
    };
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const float* B = P + kOffBody + 16 * i;
      const float* anc_row = P + kOffAnc + kBM * i;
      const int me = i * kBodySlots;
      float R[9];
#pragma unroll
      for (int e = 0; e < 9; ++e) R[e] = SLOT(me + e);
      const V3 xi = {SLOT(me + 9), SLOT(me + 10), SLOT(me + 11)};
      const V3 wi = {SLOT(me + 15), SLOT(me + 16), SLOT(me + 17)};
      const V3 ali = {SLOT(me + 18), SLOT(me + 19), SLOT(me + 20)};
      const V3 aoi = {SLOT(me + 21), SLOT(me + 22), SLOT(me + 23)};
      const V3 cl = {B[2], B[3], B[4]};
      const V3 d = {R[0] * cl.x + R[1] * cl.y + R[2] * cl.z, R[3] * cl.x + R[4] * cl.y + R[5] * cl.z,
                    R[6] * cl.x + R[7] * cl.y + R[8] * cl.z};
      const V3 com = xi + d;
      const V3 ac0 = aoi + cross(ali, d) + cross(wi, cross(wi, d));
      const float m = B[1];
      float Jx[N], Jy[N], Jz[N];
      jacobian(com, anc_row, Jx, Jy, Jz);
      const V3 f = {-m * ac0.x, -m * ac0.y, m * (-grav - ac0.z)};
      // Iw = R diag(I) R^T
      const float I0 = B[5], I1 = B[6], I2 = B[7];
      float Iw[3][3];
#pragma unroll
      for (int rr = 0; rr < 3; ++rr)
#pragma unroll
        for (int c2 = 0; c2 < 3; ++c2)
          Iw[rr][c2] = R[3 * rr] * I0 * R[3 * c2] + R[3 * rr + 1] * I1 * R[3 * c2 + 1] + R[3 * rr + 2] * I2 * R[3 * c2 + 2];
      const V3 Iw_w = {Iw[0][0] * wi.x + Iw[0][1] * wi.y + Iw[0][2] * wi.z, Iw[1][0] * wi.x + Iw[1][1] * wi.y + Iw[1][2] * wi.z,
                       Iw[2][0] * wi.x + Iw[2][1] * wi.y + Iw[2][2] * wi.z};
      const V3 Iw_a = {Iw[0][0] * ali.x + Iw[0][1] * ali.y + Iw[0][2] * ali.z,
                       Iw[1][0] * ali.x + Iw[1][1] * ali.y + Iw[1][2] * ali.z,
                       Iw[2][0] * ali.x + Iw[2][1] * ali.y + Iw[2][2] * ali.z};
      const V3 tq = (-1.f) * Iw_a - cross(wi, Iw_w);
      // angular Jacobian columns: e_c for the root rotation, a_k for the hinges on the path; Iw Jw columns
      float Wx[N], Wy[N], Wz[N];
      Wx[0] = Wx[1] = Wx[2] = 0.f; Wy[0] = Wy[1] = Wy[2] = 0.f; Wz[0] = Wz[1] = Wz[2] = 0.f;
      Wx[3] = 1.f; Wx[4] = 0.f; Wx[5] = 0.f;
      Wy[3] = 0.f; Wy[4] = 1.f; Wy[5] = 0.f;
      Wz[3] = 0.f; Wz[4] = 0.f; Wz[5] = 1.f;
#pragma unroll
      for (int k = 1; k < NB; ++k) {
        const float on = anc_row[k];
        Wx[5 + k] = on * SLOT(k * kBodySlots + 12);
        Wy[5 + k] = on * SLOT(k * kBodySlots + 13);
        Wz[5 + k] = on * SLOT(k * kBodySlots + 14);
      }
      // columns of hinges that do not carry this body are zero: skipped (uniform branches — the model is the same
      // in every lane), which leaves 8 of an ant leg's 14 columns
#pragma unroll
      for (int dd = 0; dd < N; ++dd) {
        if (dd >= 6 && anc_row[dd - 5] == 0.f) continue;
        const float ix = Iw[0][0] * Wx[dd] + Iw[0][1] * Wy[dd] + Iw[0][2] * Wz[dd];
        const float iy = Iw[1][0] * Wx[dd] + Iw[1][1] * Wy[dd] + Iw[1][2] * Wz[dd];
        const float iz = Iw[2][0] * Wx[dd] + Iw[2][1] * Wy[dd] + Iw[2][2] * Wz[dd];
#pragma unroll
        for (int e = 0; e <= dd; ++e) {
          if (e >= 6 && anc_row[e - 5] == 0.f) continue;
          M[dd][e] += m * (Jx[dd] * Jx[e] + Jy[dd] * Jy[e] + Jz[dd] * Jz[e]) + ix * Wx[e] + iy * Wy[e] + iz * Wz[e];
        }
        Q[dd] += Jx[dd] * f.x + Jy[dd] * f.y + Jz[dd] * f.z + Wx[dd] * tq.x + Wy[dd] * tq.y + Wz[dd] * tq.z;
      }
    }
    // ---- joints: armature, springs, dampers, limits, actuators ----
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const float* Jn = P + kOffJoint + 8 * j;
      M[6 + j][6 + j] += Jn[4];
      const float below = fmaxf(Jn[0] - th[j], 0.f), above = fmaxf(th[j] - Jn[1], 0.f);
      const float outside = fminf((below + above) * (1.f / kLimitRamp), 1.f);
      Q[6 + j] += tau_act[j] - Jn[2] * v[6 + j] - Jn[3] * th[j] + klim * (below - above) - clim * outside * v[6 + j];
    }
    // ---- ground contact ----
    for (int c = 0; c < ncon; ++c) {
      const float* Cn = P + kOffCon + 4 * c;
      const int body = (int)Cn[0];
      const int me = body * kBodySlots;
      const V3 lp = {Cn[1], Cn[2], Cn[3]};
      const V3 pt = {SLOT(me + 9) + SLOT(me + 0) * lp.x + SLOT(me + 1) * lp.y + SLOT(me + 2) * lp.z,
                     SLOT(me + 10) + SLOT(me + 3) * lp.x + SLOT(me + 4) * lp.y + SLOT(me + 5) * lp.z,
                     SLOT(me + 11) + SLOT(me + 6) * lp.x + SLOT(me + 7) * lp.y + SLOT(me + 8) * lp.z};
      if (pt.z < 0.f) {
        float Jx[N], Jy[N], Jz[N];
        jacobian(pt, P + kOffAnc + kBM * body, Jx, Jy, Jz);
        float vx = 0.f, vy = 0.f, vz = 0.f;
#pragma unroll
        for (int dd = 0; dd < N; ++dd) {
          vx = fmaf(Jx[dd], v[dd], vx);
          vy = fmaf(Jy[dd], v[dd], vy);
          vz = fmaf(Jz[dd], v[dd], vz);
        }
        const float depth = -pt.z;
        const float fn = fmaxf(kc * depth - cc * vz * fminf(depth * (1.f / kContactRamp), 1.f), 0.f);
        const float speed = sqrtf(vx * vx + vy * vy + 1e-12f);
        const float ft = -mu * fn * tanhf(speed * 20.f) / speed;
#pragma unroll
        for (int dd = 0; dd < N; ++dd) Q[dd] += Jx[dd] * (ft * vx) + Jy[dd] * (ft * vy) + Jz[dd] * fn;
      }
    }
    // ---- Cholesky M = G G^T in place (lower), two triangular solves ----
#pragma unroll
    for (int j = 0; j < N; ++j) {
      float dg = M[j][j];
#pragma unroll
      for (int t = 0; t < j; ++t) dg = fmaf(-M[j][t], M[j][t], dg);
      const float inv = rsqrtf(dg);
      M[j][j] = dg * inv;
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
    // ---- integrate ----
#pragma unroll
    for (int d = 0; d < N; ++d) v[d] = fminf(fmaxf(fmaf(dt, Q[d], v[d]), -vmax), vmax);
#pragma unroll
    for (int c = 0; c < 3; ++c) p[c] = fmaf(dt, v[c], p[c]);
    {
      const V3 w = {v[3], v[4], v[5]}, qv = {q[1], q[2], q[3]};
      const float qw = q[0];
      const V3 dv = qw * w + cross(w, qv);
      float nq[4] = {q[0] - 0.5f * dt * dot(w, qv), q[1] + 0.5f * dt * dv.x, q[2] + 0.5f * dt * dv.y, q[3] + 0.5f * dt * dv.z};
      const float inv = rsqrtf(nq[0] * nq[0] + nq[1] * nq[1] + nq[2] * nq[2] + nq[3] * nq[3]);
#pragma unroll
      for (int c = 0; c < 4; ++c) q[c] = nq[c] * inv;
    }
#pragma unroll
    for (int j = 0; j < J; ++j) th[j] = fmaf(dt, v[6 + j], th[j]);
  }
  if (!active) return;
  if (task) {
    bool finite = true;
#pragma unroll
    for (int c = 0; c < 3; ++c) finite = finite && fabsf(p[c]) < 3.0e38f;
#pragma unroll
    for (int c = 0; c < 4; ++c) finite = finite && fabsf(q[c]) < 3.0e38f;
#pragma unroll
    for (int j = 0; j < J; ++j) finite = finite && fabsf(th[j]) < 3.0e38f;
#pragma unroll
    for (int d = 0; d < N; ++d) finite = finite && fabsf(v[d]) < 3.0e38f;
    if (!finite) {
      // a state that left the numbers ends the episode where it was, without reward
#pragma unroll
      for (int c = 0; c < 3; ++c) p[c] = pos_io[(long long)b * 3 + c];
#pragma unroll
      for (int c = 0; c < 4; ++c) q[c] = quat_io[(long long)b * 4 + c];
#pragma unroll
      for (int j = 0; j < J; ++j) th[j] = theta_io[(long long)b * J + j];
#pragma unroll
      for (int d = 0; d < N; ++d) v[d] = 0.f;
    }
    const float* T = P + kOffTask;
    const bool healthy = p[2] > T[5] && p[2] < T[6];
    const float vx = (p[0] - x_before) / T[8];
    float reward = T[2] * vx + (healthy ? T[4] : 0.f) - T[3] * ctrl_sq;
    if (!finite) reward = 0.f;
    const bool done_now = !healthy || !finite;
    const float clip = T[7];
    float* o = obs_out + (long long)b * (1 + 4 + J + N);
    o[0] = p[2];
#pragma unroll
    for (int c = 0; c < 4; ++c) o[1 + c] = q[c];
#pragma unroll
    for (int j = 0; j < J; ++j) o[5 + j] = th[j];
#pragma unroll
    for (int d = 0; d < N; ++d) o[5 + J + d] = fminf(fmaxf(v[d], -clip), clip);
    if (!done_acc[b]) total[b] += reward;
    if (done_now) done_acc[b] = 1;
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) pos_io[(long long)b * 3 + c] = p[c];
#pragma unroll
  for (int c = 0; c < 4; ++c) quat_io[(long long)b * 4 + c] = q[c];
#pragma unroll
  for (int j = 0; j < J; ++j) theta_io[(long long)b * J + j] = th[j];
#pragma unroll
  for (int d = 0; d < N; ++d) vel_io[(long long)b * N + d] = v[d];
}

template <int NB>
int launch_spatial(const float* P, float* pos, float* quat, float* theta, float* vel, const float* ctrl, int batch, float* obs,
                   float* total, uint8_t* done, hipStream_t st) {
  const size_t lds = (size_t)NB * kBodySlots * 64 * sizeof(float);
  spatial_step_kernel<NB><<<(batch + 63) / 64, 64, lds, st>>>(P, pos, quat, theta, vel, ctrl, batch, obs, total, done);
  return last_launch_status();
}

}  // namespace

EVOX_API int evox_hip_spatial_param_floats() { return kParamFloats; }

// One environment step (the model's sub-steps) of `batch` environments in place.  params: the block laid out as above
// (ops/spatial.py builds it), on the device; pos batch x 3, quat batch x 4 (w, x, y, z), theta batch x (bodies - 1), vel
// batch x (6 + bodies - 1); ctrl batch x actuators.  obs / total / done: all null (physics only) or all set (the task's
// recipe in the kernel: observation written, the step's reward added to total[b] unless done[b] was set before the step,
// done[b] set when the episode ends).
EVOX_API int evox_hip_spatial_step(const float* params, int bodies, int contacts, int actuators, float* pos, float* quat,
                                   float* theta, float* vel, const float* ctrl, int batch, float* obs, float* total,
                                   uint8_t* done, void* stream) {
  if (bodies < 1 || bodies > kBM || contacts < 0 || contacts > kCM || actuators < 0 || actuators > kAM || batch <= 0 ||
      params == nullptr || pos == nullptr || quat == nullptr || vel == nullptr || (bodies > 1 && theta == nullptr) ||
      (actuators > 0 && ctrl == nullptr))
    return (int)hipErrorInvalidValue;
  const bool any = obs != nullptr || total != nullptr || done != nullptr;
  if (any && (obs == nullptr || total == nullptr || done == nullptr)) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  switch (bodies) {
    case 1: return launch_spatial<1>(params, pos, quat, theta, vel, ctrl, batch, obs, total, done, st);
    case 2: return launch_spatial<2>(params, pos, quat, theta, vel, ctrl, batch, obs, total, done, st);
    case 3: return launch_spatial<3>(params, pos, quat, theta, vel, ctrl, batch, obs, total, done, st);
    case 4: return launch_spatial<4>(params, pos, quat, theta, vel, ctrl, batch, obs, total, done, st);
    case 5: return launch_spatial<5>(params, pos, quat, theta, vel, ctrl, batch, obs, total, done, st);
    case 6: return launch_spatial<6>(params, pos, quat, theta, vel, ctrl, batch, obs, total, done, st);
    case 7: return launch_spatial<7>(params, pos, quat, theta, vel, ctrl, batch, obs, total, done, st);
    case 8: return launch_spatial<8>(params, pos, quat, theta, vel, ctrl, batch, obs, total, done, st);
    default: return launch_spatial<9>(params, pos, quat, theta, vel, ctrl, batch, obs, total, done, st);
  }
}
