from .eval_monitor import EvalMonitor
