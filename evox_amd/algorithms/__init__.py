from .so import *
