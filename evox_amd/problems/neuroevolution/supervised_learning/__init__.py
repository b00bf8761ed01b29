from .dataset import ArrayDataset, DatasetProblem, TensorflowDataset, synthetic_image_classification

__all__ = ["ArrayDataset", "DatasetProblem", "TensorflowDataset", "synthetic_image_classification"]
