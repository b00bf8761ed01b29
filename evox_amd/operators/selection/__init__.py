from .misc import RouletteWheelSelection, TopkFit, UniformRand, select_rand_pbest, topk_fit, uniform_rand
from .non_dominate import (NonDominate, crowding_distance, crowding_distance_sort, non_dominate,
                           non_dominated_sort)
from .rvea_selection import ReferenceVectorGuided, ref_vec_guided
from .tournament import Tournament, tournament_multi_fit, tournament_single_fit
