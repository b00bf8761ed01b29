from .differential_evolution import (DifferentialEvolve, de_arith_recom, de_bin_cross, de_diff_sum, de_exp_cross,
                                     differential_evolve)
from .sbx import SimulatedBinary, simulated_binary
from .paired_sbx import SBXCrossover, sbx
from .uniform import OnePoint, UniformRand, one_point, uniform_rand
