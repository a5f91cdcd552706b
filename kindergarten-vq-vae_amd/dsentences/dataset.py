"""dSentences map-style dataset (counterpart of datasets/dSentences/dSentencesDataset.py:13-64).

Same constructor and item layout as the reference class.  The package is not called `datasets` on purpose:
that name would shadow the HuggingFace `datasets` package that `transformers` probes for."""
from typing import Union

import numpy as np
from torch import Tensor, as_tensor
from torch.utils.data import Dataset


class dSentencesDataset(Dataset):
    def __init__(self, sentences_path: str, latent_classes_labels_path: str = None,
                 latent_classes_one_hot_path: str = None):
        self.sentences = np.load(sentences_path).tolist()
        self.latent_classes_labels = None
        self.latent_classes_one_hot = None
        if latent_classes_labels_path is not None and latent_classes_one_hot_path is not None:
            self.latent_classes_labels: Tensor = as_tensor(np.load(latent_classes_labels_path)).long()
            self.latent_classes_one_hot: Tensor = as_tensor(np.load(latent_classes_one_hot_path))
            n = len(self.sentences)
            if n != self.latent_classes_labels.shape[0]:
                raise AssertionError(f"Provided {n} sentences but {self.latent_classes_labels.shape[0]} latent classes labels.")
            if n != self.latent_classes_one_hot.shape[0]:
                raise AssertionError(f"Provided {n} sentences but {self.latent_classes_one_hot.shape[0]} latent classes one-hot labels.")

    def __len__(self) -> int:
        return len(self.sentences)

    def __getitem__(self, idx) -> Union[str, dict]:
        if self.latent_classes_labels is None:
            return {"sentence": self.sentences[idx]}
        return {"sentence": self.sentences[idx],
                "latent_classes_labels": self.latent_classes_labels[idx],
                "latent_classes_one_hot": self.latent_classes_one_hot[idx]}
