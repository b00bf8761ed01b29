from .clustered_algorithm import ClusterdAlgorithm, RandomMaskAlgorithm
from .coevolution import Coevolution, VectorizedCoevolution
from .tree_algorithm import TreeAlgorithm

__all__ = ["ClusterdAlgorithm", "RandomMaskAlgorithm", "Coevolution", "VectorizedCoevolution", "TreeAlgorithm"]
