from .mo import *
from .so import *
