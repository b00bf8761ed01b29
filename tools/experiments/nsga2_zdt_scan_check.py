import os, sys, torch
sys.path.insert(0, "/root/repo")
import evox_amd as ex
from evox_amd import algorithms, problems, workflows, metrics
dev = torch.device("cuda")
def run(flag):
    os.environ["EVOX_AMD_NDS_M2_SCAN"] = flag
    prob = problems.numerical.ZDT1(n=30)
    algo = algorithms.NSGA2(lb=torch.zeros(30, device=dev), ub=torch.ones(30, device=dev), n_objs=2, pop_size=4000)
    wf = workflows.StdWorkflow(algo, prob)
    s = wf.init(ex.PRNGKey(3).to(dev))
    for _ in range(60): s = wf.step(s)
    st = s.get_child_state("algorithm")
    return st.population.clone(), st.fitness.clone(), float(metrics.igd(st.fitness, prob.pf().to(dev)))
p1, f1, i1 = run("1"); p0, f0, i0 = run("0")
print("identical populations:", torch.equal(p1, p0), "igd", i1, i0)
