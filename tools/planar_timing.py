"""Time per environment step of the planar physics tasks for a batch of environments on the GPU.
usage: planar_timing.py [batch]"""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evox_amd as ex
from evox_amd.problems.neuroevolution.reinforcement_learning import make_tensor_env

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda")
for name in ("inverted_pendulum", "reacher", "swimmer", "hopper", "walker2d", "halfcheetah"):
    env = make_tensor_env(name)
    st, o = env.reset(ex.PRNGKey(0).to(dev), batch, dev)
    a = torch.zeros(batch, env.act_dim, device=dev)
    for _ in range(3):
        st, o, r, d = env.step(st, a)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        st, o, r, d = env.step(st, a)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    print(json.dumps({"task": name, "batch": batch, "ms_per_env_step": round(ms, 4), "substeps": env.engine.model.substeps,
                      "env_steps_per_s": round(batch / ms * 1e3)}), flush=True)
