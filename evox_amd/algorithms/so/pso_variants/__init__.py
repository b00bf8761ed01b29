from .cso import CSO, CSOState
from .pso import PSO
from .variants import CLPSO, DMSPSOEL, FIPS, FSPSO, SLPSOGS, SLPSOUS, SwmmPSO
