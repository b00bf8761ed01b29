from . import crossover, gaussian_process, mutation, sampling, selection
from .selection import crowding_distance, crowding_distance_sort, non_dominated_sort
