"""Per-frame image feature cache (interface of deva/inference/image_feature_store.py:7-48)."""
import warnings
from typing import Iterable, Tuple

import torch


class ImageFeatureStore:
    """Caches (multi-scale features, key feature, key, shrinkage, selection) per frame index so the
    encoder runs once per frame even when several consumers need it.  Callers must `delete`."""

    def __init__(self, network, no_warning: bool = False):
        self.network = network
        self._store = {}
        self.no_warning = no_warning

    def _encode_feature(self, index: int, image: torch.Tensor) -> None:
        ms_features, feat = self.network.encode_image(image)
        key, shrinkage, selection = self.network.transform_key(feat)
        self._store[index] = (ms_features, feat, key, shrinkage, selection)

    def get_ms_features(self, index, image) -> Iterable[torch.Tensor]:
        if index not in self._store:
            self._encode_feature(index, image)
        return self._store[index][0]

    def get_key(self, index, image) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        if index not in self._store:
            self._encode_feature(index, image)
        return self._store[index][2:]

    def delete(self, index) -> None:
        self._store.pop(index, None)

    def __len__(self):
        return len(self._store)

    def __del__(self):
        if len(self._store) > 0 and not self.no_warning:
            warnings.warn(f'Leaking {self._store.keys()} in the image feature store')
