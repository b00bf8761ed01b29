from .adaptive import SHADE, CoDE, JaDE, SaDE
from .de import DE, ODE
