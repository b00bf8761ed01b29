from . import numerical
