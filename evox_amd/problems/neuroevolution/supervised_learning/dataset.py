"""Supervised-learning fitness: the loss of every individual on the next mini-batch.
Reference: ``src/evox/problems/neuroevolution/supervised_learning/tfds.py:26-136`` (TensorFlow
Datasets + grain loader behind an ``io_callback``).

``DatasetProblem`` keeps the whole dataset as tensors on the evaluation device (datasets of
many GB fit next to the population in 288 GB of HBM) and draws reproducible shuffled batches
from a Philox permutation per epoch – no host loader, no host↔device traffic per step.
``TensorflowDataset`` keeps the reference's signature: it materialises a TFDS split into an
``ArrayDataset`` when ``tensorflow_datasets`` is importable, and otherwise reads the dataset's
files from disk (numpy archives, IDX files, CIFAR-10 binary batches, parquet)."""
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


# ---------------------------------------------------------------------------------------------------------------------
# datasets from files on disk (what a TFDS download leaves behind, or what other tools export)
# ---------------------------------------------------------------------------------------------------------------------
def _read_idx(path):
    """An IDX file (the MNIST family's format: magic, dimensions, big-endian payload), optionally gzip-compressed."""
    import gzip
    import struct

    import numpy as np

    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "rb") as f:
        raw = f.read()
    zero, code, ndim = struct.unpack(">HBB", raw[:4])
    if zero != 0:
        raise ValueError(f"{path}: not an IDX file")
    dtype = {0x08: ">u1", 0x09: ">i1", 0x0B: ">i2", 0x0C: ">i4", 0x0D: ">f4", 0x0E: ">f8"}[code]
    shape = struct.unpack(">" + "I" * ndim, raw[4:4 + 4 * ndim])
    return np.frombuffer(raw, dtype=dtype, offset=4 + 4 * ndim).reshape(shape).astype(dtype[1:])


def _first(paths):
    import os

    return next((p for p in paths if os.path.exists(p)), None)


def load_local_dataset(name: str, split: str = "train", datadir: Optional[str] = None) -> Optional["ArrayDataset"]:
    """A dataset from files under ``datadir`` (default ``~/tensorflow_datasets``), or None when nothing is found there.
    Looked for, in this order, under ``<datadir>/<name>/`` (archives named after the dataset also in ``<datadir>/`` itself):

    * ``<split>.npz`` / ``<name>_<split>.npz`` / ``<name>.npz`` — numpy archives; every array of the split's length
      becomes a column (``x_train`` / ``y_train`` style names are mapped to ``image`` / ``label``);
    * IDX files (``train-images-idx3-ubyte[.gz]`` + ``train-labels-idx1-ubyte[.gz]``, ``t10k-…`` for the test split):
      the MNIST family as distributed;
    * CIFAR-10's binary batches (``data_batch_*.bin`` / ``test_batch.bin``: one label byte and 3072 pixel bytes a row);
    * ``<split>*.parquet`` — columns of numbers or of fixed-length lists (read with pyarrow).

    Images come back as ``uint8`` in NHWC layout, like the TFDS feature ``image``."""
    import glob
    import os

    import numpy as np

    root = os.path.expanduser(datadir or "~/tensorflow_datasets")
    own = os.path.join(root, name)
    test = split.startswith("test")
    for d in (own, root):
        # in <datadir> itself only archives that carry the dataset's name count; everything else must sit in <datadir>/<name>/
        npz = _first(([os.path.join(d, f"{split}.npz")] if d == own else [])
                     + [os.path.join(d, f"{name}_{split}.npz"), os.path.join(d, f"{name}.npz")])
        if npz is not None:
            with np.load(npz, allow_pickle=False) as z:
                arrays = {k: z[k] for k in z.files}
            tag = "test" if test else "train"
            tagged = {k: v for k, v in arrays.items() if k.endswith("_" + tag)}
            if tagged:                                       # Keras-style archive holding both splits
                arrays = {k[:-len(tag) - 1]: v for k, v in tagged.items()}
            rename = {"x": "image", "y": "label", "images": "image", "labels": "label"}
            arrays = {rename.get(k, k): v for k, v in arrays.items()}
            n = max(v.shape[0] for v in arrays.values())
            arrays = {k: v for k, v in arrays.items() if v.shape[0] == n}
            if "image" in arrays and arrays["image"].ndim == 3:
                arrays["image"] = arrays["image"][..., None]
            return ArrayDataset({k: torch.as_tensor(np.ascontiguousarray(v)) for k, v in arrays.items()})
        if d != own:
            break
        stem = "t10k" if test else "train"
        img = _first([os.path.join(d, f"{stem}-images-idx3-ubyte{e}") for e in ("", ".gz")]
                     + [os.path.join(d, f"{stem}-images.idx3-ubyte{e}") for e in ("", ".gz")])
        lab = _first([os.path.join(d, f"{stem}-labels-idx1-ubyte{e}") for e in ("", ".gz")]
                     + [os.path.join(d, f"{stem}-labels.idx1-ubyte{e}") for e in ("", ".gz")])
        if img is not None and lab is not None:
            x, y = _read_idx(img), _read_idx(lab)
            return ArrayDataset({"image": torch.as_tensor(x[..., None] if x.ndim == 3 else x),
                                 "label": torch.as_tensor(y.astype(np.int64))})
        bins = sorted(glob.glob(os.path.join(d, "**", "test_batch.bin" if test else "data_batch_*.bin"), recursive=True))
        if bins:
            rows = np.concatenate([np.fromfile(b, dtype=np.uint8).reshape(-1, 3073) for b in bins])
            return ArrayDataset({"image": torch.as_tensor(np.ascontiguousarray(rows[:, 1:].reshape(-1, 3, 32, 32).transpose(0, 2, 3, 1))),
                                 "label": torch.as_tensor(rows[:, 0].astype(np.int64))})
        pq = sorted(glob.glob(os.path.join(d, f"{split}*.parquet")))
        if pq:
            import pyarrow.parquet as papq

            table = papq.read_table(pq[0]) if len(pq) == 1 else __import__("pyarrow").concat_tables([papq.read_table(f) for f in pq])
            arrays = {}
            for col in table.column_names:
                try:
                    arrays[col] = torch.as_tensor(np.stack(table[col].to_numpy(zero_copy_only=False)))
                except (ValueError, TypeError):
                    continue                                 # a column that is not numeric (strings, encoded images)
            if arrays:
                return ArrayDataset(arrays)
    return None


class TensorflowDataset(DatasetProblem):
    """The reference's signature (``tfds.py:26-62``).  The split is materialised once into device tensors: through
    ``tensorflow_datasets`` when that package is importable, otherwise from the files a download left on disk or other
    tools exported (:func:`load_local_dataset`: numpy archives, IDX files, CIFAR-10 binary batches, parquet)."""

    def __init__(self, dataset: str, batch_size: int, loss_func: Callable, split: str = "train",
                 operations: Optional[List[Any]] = None, datadir: Optional[str] = None, seed: int = 0,
                 try_gcs: bool = True, device=None, batched_loss: bool = False):
        try:
            import tensorflow_datasets as tfds
        except ImportError as e:
            local = load_local_dataset(dataset, split, datadir)
            if local is None:
                raise ImportError(f'TensorflowDataset found neither tensorflow-datasets ("{e}") nor files of "{dataset}" '
                                  f'under {datadir or "~/tensorflow_datasets"} (npz, IDX, CIFAR-10 binary or parquet); use '
                                  "DatasetProblem with an ArrayDataset for in-memory data")
            super().__init__(local.to(device) if device is not None else local, batch_size, loss_func, seed, operations,
                             batched_loss)
            return
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
