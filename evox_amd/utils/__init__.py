from .common import (AggregationFunction, TreeAndVector, cal_max, chebyshev_dist, compose, cos_dist,
                     dominate_relation, euclidean_dist, frames2gif, manhattan_dist, min_by, new_dist_mat, pair_max,
                     pairwise_chebyshev_dist, pairwise_euclidean_dist, pairwise_func, pairwise_manhattan_dist,
                     parse_opt_direction, rank, rank_based_fitness)
from .io import to_x32_if_needed, x32_func_call
from . import optimizers
from .optimizers import ClipUp, OptaxWrapper, apply_updates
