from .basic import (Ackley, Griewank, Rastrigin, Rosenbrock, Schwefel, Sphere, ackley_func, griewank_func,
                    rastrigin_func, rosenbrock_func, schwefel_func, sphere_func)
from .dtlz import DTLZ1, DTLZ2, DTLZ3, DTLZ4, DTLZ5, DTLZ6, DTLZ7, DTLZTestSuit
from .zdt import ZDT1, ZDT2, ZDT3, ZDT4, ZDT6, ZDTTestSuit
