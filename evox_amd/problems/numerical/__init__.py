from .basic import (Ackley, Ellipsoid, Griewank, Rastrigin, Rosenbrock, Schwefel, Sphere, ackley_func, griewank_func,
                    rastrigin_func, rosenbrock_func, schwefel_func, sphere_func)
from .dtlz import DTLZ1, DTLZ2, DTLZ3, DTLZ4, DTLZ5, DTLZ6, DTLZ7, DTLZTestSuit
from .zdt import ZDT1, ZDT2, ZDT3, ZDT4, ZDT6, ZDTTestSuit
from .cec2022 import (CEC2022TestSuit, F1_CEC2022, F2_CEC2022, F3_CEC2022, F4_CEC2022, F5_CEC2022, F6_CEC2022,
                      F7_CEC2022, F8_CEC2022, F9_CEC2022, F10_CEC2022, F11_CEC2022, F12_CEC2022)
from .lsmop import LSMOP1, LSMOP2, LSMOP3, LSMOP4, LSMOP5, LSMOP6, LSMOP7, LSMOP8, LSMOP9
from .maf import (MaF1, MaF2, MaF3, MaF4, MaF5, MaF6, MaF7, MaF8, MaF9, MaF10, MaF11, MaF12, MaF13, MaF14, MaF15)
