"""Pack the official CEC 2022 single-objective benchmark constants (shift vectors, rotation
matrices, shuffle permutations; plain text, one file per function and dimension) into the one
compressed archive that ``evox_amd.problems.numerical.cec2022`` loads.

usage: python tools/pack_cec2022.py <dir with shift_data_*.txt, M_*_D*.txt, shuffle_data_*_D*.txt> [out.npz]"""
import sys
from pathlib import Path

import numpy as np


def main(src, out):
    src = Path(src)
    arrays = {}
    for f in range(1, 13):
        arrays[f"shift_{f}"] = np.atleast_2d(np.loadtxt(src / f"shift_data_{f}.txt", dtype=np.float64))
        for d in (2, 10, 20):
            arrays[f"rot_{f}_{d}"] = np.loadtxt(src / f"M_{f}_D{d}.txt", dtype=np.float64).reshape(-1, d)
            p = src / f"shuffle_data_{f}_D{d}.txt"
            if p.exists():
                arrays[f"shuffle_{f}_{d}"] = np.loadtxt(p, dtype=np.int64).reshape(-1)
    np.savez_compressed(out, **arrays)
    print(f"wrote {out}: {len(arrays)} arrays")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "evox_amd/problems/numerical/data/cec2022.npz")
