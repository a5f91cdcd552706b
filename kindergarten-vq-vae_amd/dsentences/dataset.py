"""dSentences map-style dataset: same constructor arguments and item layout as the reference's
datasets/dSentences/dSentencesDataset.py:13-64 (sentences .npy + optional factor labels / one-hot .npy).

The package is deliberately not called `datasets`: that name would shadow the HuggingFace `datasets` package that
`transformers` probes for at import time."""
from typing import Optional

import numpy as np
import torch
from torch.utils.data import Dataset


def _load_tensor(path: Optional[str], dtype=None):
    if path is None:
        return None
    t = torch.as_tensor(np.load(path))
    return t.to(dtype) if dtype is not None else t


class dSentencesDataset(Dataset):
    """item = {"sentence": str} or {"sentence", "latent_classes_labels" (int64 [9]), "latent_classes_one_hot" (float [42])}"""

    def __init__(self, sentences_path: str, latent_classes_labels_path: str = None, latent_classes_one_hot_path: str = None):
        self.sentences = np.load(sentences_path).tolist()
        both = latent_classes_labels_path is not None and latent_classes_one_hot_path is not None
        self.latent_classes_labels = _load_tensor(latent_classes_labels_path, torch.int64) if both else None
        self.latent_classes_one_hot = _load_tensor(latent_classes_one_hot_path) if both else None
        for what, t in (("latent classes labels", self.latent_classes_labels),
                        ("latent classes one-hot labels", self.latent_classes_one_hot)):
            if t is not None and t.shape[0] != len(self.sentences):
                raise AssertionError(f"Provided {len(self.sentences)} sentences but {t.shape[0]} {what}: "
                                     f"every sentence needs its {what}")

    def __len__(self) -> int:
        return len(self.sentences)

    def __getitem__(self, idx) -> dict:
        item = {"sentence": self.sentences[idx]}
        if self.latent_classes_labels is not None:
            item["latent_classes_labels"] = self.latent_classes_labels[idx]
            item["latent_classes_one_hot"] = self.latent_classes_one_hot[idx]
        return item
