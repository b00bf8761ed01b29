"""Generation time of every multi-objective algorithm on DTLZ2 (m=3) on the GPU.
usage: mo_survey.py [pop] [NAME ...] ; one JSON line per algorithm (time per generation, IGD after 65 generations)."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evox_amd as ex
from evox_amd import algorithms, problems, workflows

pop = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
names = ["NSGA2", "NSGA3", "RVEA", "RVEAa", "MOEAD", "MOEADDRA", "MOEADM2M", "EAGMOEAD", "IBEA", "BCEIBEA", "SPEA2", "HypE",
         "KnEA", "BiGE", "GDE3", "SRA", "TDEA", "LMOCSO", "IMMOEA"]
if len(sys.argv) > 2:
    names = [n for n in names if n in sys.argv[2:]]
dev = torch.device("cuda")
prob = problems.numerical.DTLZ2(m=3)
lb, ub = torch.zeros(prob.d, device=dev), torch.ones(prob.d, device=dev)
# the code of a kernel is loaded at its first launch (milliseconds): launch the selection loops once on throw-away data, so
# that a loop whose first use falls into the timed generations (SPEA2's first truncation) is not charged for it
from evox_amd.ops import moselect as _sel
_w = torch.rand(max(2 * pop, 4096), 3, device=dev)
_f, _c = _sel.ibea_fitness(_w, 0.05)
_sel.ibea_remove_streaming(_w, _f, _c, 0.05, _w.shape[0] // 2)
_sel.spea2_truncate(_w, torch.ones(_w.shape[0], dtype=torch.bool, device=dev), _w.shape[0] // 2)
torch.cuda.synchronize()
for name in names:
    row = {"algorithm": name, "pop": pop}
    try:
        algo = algorithms.LMOCSO(3, lb, ub, pop) if name == "LMOCSO" else getattr(algorithms, name)(lb, ub, 3, pop)
        wf = workflows.StdWorkflow(algo, prob)
        state = wf.init(ex.PRNGKey(0).to(dev))
        for _ in range(5):
            state = wf.step(state)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        steps = 20
        for _ in range(steps):
            state = wf.step(state)
        torch.cuda.synchronize()
        row["ms_per_generation"] = (time.perf_counter() - t0) / steps * 1e3
        # quality after 25 + 40 more generations: IGD to the true front (finite and small means selection works)
        for _ in range(40):
            state = wf.step(state)
        from evox_amd import metrics
        fit = state.get_child_state("algorithm").fitness
        fit = fit[torch.isfinite(fit).all(1)]
        row["igd_after_65_generations"] = float(metrics.igd(fit, prob.pf().to(dev)))
        row["n_finite"] = int(fit.shape[0])
    except Exception as e:                                            # noqa: BLE001
        row["error"] = f"{type(e).__name__}: {e}"[:160]
    print(json.dumps(row), flush=True)
