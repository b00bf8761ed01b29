from .de_variants import *
from .es_variants import *
from .pso_variants import *
