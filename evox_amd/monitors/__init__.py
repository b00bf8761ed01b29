from .eval_monitor import EvalMonitor
from .pop_monitor import PopMonitor
from .std_monitors import StdMOMonitor, StdSOMonitor


def __getattr__(name):
    # pyarrow is only needed by the EvoXVis writer
    if name == "EvoXVisMonitor":
        from .evoxvis_monitor import EvoXVisMonitor
        return EvoXVisMonitor
    raise AttributeError(name)


__all__ = ["EvalMonitor", "PopMonitor", "StdSOMonitor", "StdMOMonitor", "EvoXVisMonitor"]
