from .reinforcement_learning import *
from .supervised_learning import *
