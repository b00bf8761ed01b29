from .pso import PSO
