from .basic import (Ackley, Griewank, Rastrigin, Rosenbrock, Schwefel, Sphere, ackley_func, griewank_func,
                    rastrigin_func, rosenbrock_func, schwefel_func, sphere_func)
