"""Policy / network models for neuroevolution.

``MLP`` is a parameter-vector network: a whole population lives in one ``(pop, n_params)``
tensor (bf16 or fp32, typically tens of GB on a 288 GB device) and the forward pass of every
individual on its own observations is a strided batched GEMM per layer straight out of that
tensor – no per-individual modules, no parameter copies."""
from .mlp import MLP, batched_mlp_forward, mlp_param_count

__all__ = ["MLP", "batched_mlp_forward", "mlp_param_count"]
