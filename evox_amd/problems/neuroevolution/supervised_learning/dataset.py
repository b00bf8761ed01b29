"""Supervised-learning fitness: the loss of every individual on the next mini-batch.
Reference: ``src/evox/problems/neuroevolution/supervised_learning/tfds.py:26-136`` (TensorFlow
Datasets + grain loader behind an ``io_callback``).

``DatasetProblem`` keeps the whole dataset as tensors on the evaluation device (datasets of
many GB fit next to the population in 288 GB of HBM) and draws reproducible shuffled batches
from a Philox permutation per epoch – no host loader, no host↔device traffic per step.
``TensorflowDataset`` keeps the reference's signature: it materialises a TFDS split into an
``ArrayDataset`` when ``tensorflow_datasets`` is importable and raises ``ImportError``
otherwise."""
from __future__ import annotations

from typing import Any, Callable, Dict, List, Optional

import torch

from ....core import random as rnd
from ....core.problem import Problem
from ....core.state import State


class ArrayDataset:
    """A dict of equally long tensors, e.g. ``{"image": (N,28,28,1), "label": (N,)}``."""

    def __init__(self, arrays: Dict[str, torch.Tensor], device=None):
        n = {int(v.shape[0]) for v in arrays.values()}
        if len(n) != 1:
            raise ValueError("all arrays must share their first dimension")
        self.arrays = {k: (torch.as_tensor(v).to(device) if device is not None else torch.as_tensor(v))
                       for k, v in arrays.items()}
        self.size = n.pop()

    def to(self, device):
        return ArrayDataset({k: v.to(device) for k, v in self.arrays.items()})

    def take(self, index):
        return {k: v[index.to(v.device)] for k, v in self.arrays.items()}


def synthetic_image_classification(key, n=2048, classes=10, shape=(28, 28, 1), noise=0.35):
    """Class-conditional blobs: each class has a fixed random template, samples are the
    template plus noise, scaled to 0..255 like image bytes."""
    k_t, k_l, k_n = rnd.split(key, 3)
    d = 1
    for s in shape:
        d *= s
    templates = rnd.uniform(k_t, (classes, d))
    labels = rnd.randint(k_l, (n,), 0, classes)
    x = (templates[labels] + noise * rnd.normal(k_n, (n, d))).clamp(0, 1) * 255.0
    return ArrayDataset({"image": x.reshape(n, *shape), "label": labels})


class DatasetProblem(Problem):
    """``loss_func(weights, data) -> loss`` is written for ONE individual and vmapped over the
    population with ``torch.func.vmap`` (set ``batched_loss=True`` when it already handles a
    leading population dimension, e.g. through ``evox_amd.models.MLP.batched_apply``)."""

    def __init__(self, dataset: ArrayDataset, batch_size: int, loss_func: Callable, seed: int = 0,
                 operations: Optional[List[Callable]] = None, batched_loss: bool = False):
        super().__init__()
        self.dataset, self.batch_size, self.loss_func = dataset, batch_size, loss_func
        self.seed, self.operations, self.batched_loss = seed, list(operations or []), batched_loss
        self.batches_per_epoch = dataset.size // batch_size
        if self.batches_per_epoch < 1:
            raise ValueError("batch_size exceeds the dataset size")

    def setup(self, key):
        return State(step=0, data_key=rnd.PRNGKey(self.seed))

    def _batch(self, state):
        epoch, pos = divmod(state.step, self.batches_per_epoch)
        perm = rnd.permutation(rnd.fold_in(state.data_key, epoch), self.dataset.size)
        data = self.dataset.take(perm[pos * self.batch_size:(pos + 1) * self.batch_size])
        for op in self.operations:
            data = op(data)
        return data

    def evaluate(self, state, pop):
        data = self._batch(state)
        from ....core.tree import tree_leaves, tree_map
        device = tree_leaves(pop)[0].device
        data = tree_map(lambda t: t.to(device), data)
        if self.batched_loss:
            loss = self.loss_func(pop, data)
        else:
            loss = torch.func.vmap(self.loss_func, in_dims=(0, None))(pop, data)
        return loss, state.replace(step=state.step + 1)


class TensorflowDataset(DatasetProblem):
    def __init__(self, dataset: str, batch_size: int, loss_func: Callable, split: str = "train",
                 operations: Optional[List[Any]] = None, datadir: Optional[str] = None, seed: int = 0,
                 try_gcs: bool = True, device=None, batched_loss: bool = False):
        try:
            import tensorflow_datasets as tfds
        except ImportError as e:
            raise ImportError(f'TensorflowDataset requires tensorflow-datasets but got "{e}" when importing; '
                              "use DatasetProblem with an ArrayDataset for in-memory data")
        kwargs = dict(split=split, try_gcs=try_gcs)
        if datadir is not None:
            kwargs["data_dir"] = datadir
        source = tfds.data_source(dataset, **kwargs)
        import numpy as np
        columns: Dict[str, list] = {}
        for i in range(len(source)):
            for k, v in source[i].items():
                if hasattr(v, "dtype") or isinstance(v, (int, float)):
                    columns.setdefault(k, []).append(np.asarray(v))
        arrays = {k: torch.as_tensor(np.stack(v)) for k, v in columns.items()}
        super().__init__(ArrayDataset(arrays, device), batch_size, loss_func, seed, operations, batched_loss)
