// A small native vectorised environment pool: N independent copies of a classic-control task
// stepped together (OpenMP over environments), with structure-of-arrays state.  It plays the role
// the external `envpool` C++ library plays for the reference's EnvPool problem.  An environment
// that has terminated or been truncated stays frozen (reward 0) until the next reset.
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <vector>

namespace {

enum EnvId { kCartPole = 0, kPendulum = 1 };

struct SplitMix {            // per-environment seeding stream
  uint64_t s;
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  double uniform(double lo, double hi) { return lo + (hi - lo) * ((next() >> 11) * (1.0 / 9007199254740992.0)); }
};

struct Pool {
  int env_id, n, max_steps, obs_dim, act_dim;
  uint64_t seed = 0;
  std::vector<double> st;          // n x 4 state
  std::vector<int> t;
  std::vector<uint8_t> done;
};

inline void write_obs(const Pool& p, int i, float* obs) {
  const double* s = &p.st[(size_t)i * 4];
  if (p.env_id == kCartPole) {
    for (int k = 0; k < 4; ++k) obs[(size_t)i * 4 + k] = (float)s[k];
  } else {
    obs[(size_t)i * 3 + 0] = (float)cos(s[0]);
    obs[(size_t)i * 3 + 1] = (float)sin(s[0]);
    obs[(size_t)i * 3 + 2] = (float)s[1];
  }
}

}  // namespace

extern "C" {

void* evox_cpu_envpool_create(int env_id, int num_envs, int max_steps) {
  if (env_id != kCartPole && env_id != kPendulum) return nullptr;
  Pool* p = new Pool();
  p->env_id = env_id; p->n = num_envs; p->max_steps = max_steps;
  p->obs_dim = env_id == kCartPole ? 4 : 3;
  p->act_dim = 1;
  p->st.assign((size_t)num_envs * 4, 0.0);
  p->t.assign(num_envs, 0);
  p->done.assign(num_envs, 0);
  return p;
}

void evox_cpu_envpool_destroy(void* h) { delete static_cast<Pool*>(h); }
int evox_cpu_envpool_obs_dim(void* h) { return static_cast<Pool*>(h)->obs_dim; }
int evox_cpu_envpool_act_dim(void* h) { return static_cast<Pool*>(h)->act_dim; }
void evox_cpu_envpool_seed(void* h, uint64_t seed) { static_cast<Pool*>(h)->seed = seed; }

void evox_cpu_envpool_reset(void* h, float* obs) {
  Pool& p = *static_cast<Pool*>(h);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < p.n; ++i) {
    SplitMix rng{p.seed * 0x2545F4914F6CDD1Dull + (uint64_t)i};
    double* s = &p.st[(size_t)i * 4];
    if (p.env_id == kCartPole) {
      for (int k = 0; k < 4; ++k) s[k] = rng.uniform(-0.05, 0.05);
    } else {
      s[0] = rng.uniform(-M_PI, M_PI); s[1] = rng.uniform(-1.0, 1.0); s[2] = s[3] = 0.0;
    }
    p.t[i] = 0; p.done[i] = 0;
    write_obs(p, i, obs);
  }
}

// actions: n floats (CartPole: push right when > 0.5; Pendulum: torque, clipped to [-2, 2]).
void evox_cpu_envpool_step(void* h, const float* actions, float* obs, float* reward, uint8_t* terminated,
                           uint8_t* truncated) {
  Pool& p = *static_cast<Pool*>(h);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < p.n; ++i) {
    double* s = &p.st[(size_t)i * 4];
    if (p.done[i]) { reward[i] = 0.f; write_obs(p, i, obs); continue; }   // flags keep their last value
    bool term = false;
    if (p.env_id == kCartPole) {
      const double force = actions[i] > 0.5f ? 10.0 : -10.0;
      const double g = 9.8, mc = 1.0, mp = 0.1, l = 0.5, tau = 0.02, total = mc + mp, pml = mp * l;
      const double c = cos(s[2]), sn = sin(s[2]);
      const double temp = (force + pml * s[3] * s[3] * sn) / total;
      const double th_acc = (g * sn - c * temp) / (l * (4.0 / 3.0 - mp * c * c / total));
      const double x_acc = temp - pml * th_acc * c / total;
      const double x = s[0] + tau * s[1], xd = s[1] + tau * x_acc, th = s[2] + tau * s[3], thd = s[3] + tau * th_acc;
      s[0] = x; s[1] = xd; s[2] = th; s[3] = thd;
      term = fabs(x) > 2.4 || fabs(th) > 12.0 * 2.0 * M_PI / 360.0;
      reward[i] = 1.f;
    } else {
      double u = actions[i]; u = u < -2.0 ? -2.0 : (u > 2.0 ? 2.0 : u);
      const double g = 10.0, m = 1.0, l = 1.0, dt = 0.05;
      double nth = fmod(s[0] + M_PI, 2.0 * M_PI); if (nth < 0) nth += 2.0 * M_PI; nth -= M_PI;
      reward[i] = (float)-(nth * nth + 0.1 * s[1] * s[1] + 0.001 * u * u);
      double thd = s[1] + (3.0 * g / (2.0 * l) * sin(s[0]) + 3.0 / (m * l * l) * u) * dt;
      thd = thd < -8.0 ? -8.0 : (thd > 8.0 ? 8.0 : thd);
      s[1] = thd; s[0] += thd * dt;
    }
    p.t[i] += 1;
    const bool trunc = p.t[i] >= p.max_steps;
    terminated[i] = term; truncated[i] = trunc;
    p.done[i] = term || trunc;
    write_obs(p, i, obs);
  }
}

}  // extern "C"
