from .batched_cma import BatchedCMAES
from .cma_es import BIPOPCMAES, CMAES, IPOPCMAES, SepCMAES
from .es_family import (ARS, ASEBO, DES, ESMC, PGPE, SNES, XNES, GuidedES, Noise_reuse_es, OpenES, PersistentES,
                        SeparableNES)
from .ma_es import LMMAES, MAES, AMaLGaM, IndependentAMaLGaM
from .structured import CR_FM_NES, LES, RMES, init_les_params
