"""Pre-defined problems: numerical benchmarks, neuroevolution tasks, EvoXBench adapters."""
from . import evoxbench, neuroevolution, numerical

__all__ = ["numerical", "neuroevolution", "evoxbench"]
