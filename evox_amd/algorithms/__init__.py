from .mo import *
from .so import *
from .containers import *
from . import containers, mo, so
