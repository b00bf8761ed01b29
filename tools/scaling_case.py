"""One case of ``tools/scaling_sweep.py`` run for a fixed number of generations, for a kernel trace:

    rocprofv3 --kernel-trace --stats -d gpurun_out/prof -- python3 tools/scaling_case.py shade 1000000 100 [generations]
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import torch

import scaling_sweep


def main():
    fam, pop, dim = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    gens = int(sys.argv[4]) if len(sys.argv) > 4 else 30
    make, _ = scaling_sweep.cases(torch.device("cuda:0"))[fam]
    wf, key = make(pop, dim)
    state = wf.init(key)
    for _ in range(gens):
        state = wf.step(state)
    torch.cuda.synchronize()
    print(f"{fam} pop={pop} dim={dim}: {gens} generations", flush=True)


if __name__ == "__main__":
    main()
