"""Data formats on the input side of the hot path (`from datasets import TUDataset`, main_rna_puzzles.py:12,
inference_rna_puzzles.py:9) plus a minimal batch loader standing in for torch_geometric's DataLoader.
The QM9 reader of the reference needs RDKit + the raw SDF download and is out of scope."""
from .tu_dataset import DataLoader, TUDataset, read_tu_data

__all__ = ["TUDataset", "DataLoader", "read_tu_data"]
