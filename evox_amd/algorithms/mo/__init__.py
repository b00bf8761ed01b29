from .bce_ibea import BCEIBEA
from .im_moea import IMMOEA
from .indicator import IBEA, SPEA2, HypE
from .knea import KnEA, BiGE, TDEA
from .moead import MOEAD, MOEADDRA
from .moead_variants import EAGMOEAD, MOEADM2M
from .nsga2 import GDE3, NSGA2
from .nsga3 import NSGA3
from .rvea import RVEA, RVEAa
from .sra import SRA, LMOCSO

__all__ = ["BCEIBEA", "BiGE", "EAGMOEAD", "GDE3", "HypE", "IBEA", "IMMOEA", "KnEA", "LMOCSO", "MOEAD", "MOEADDRA",
           "MOEADM2M", "NSGA2", "NSGA3", "RVEA", "RVEAa", "SPEA2", "SRA", "TDEA"]
