from .pso_variants import *
