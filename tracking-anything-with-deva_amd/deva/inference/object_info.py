"""Per-object metadata (interface of deva/inference/object_info.py:7-62)."""
from collections import Counter
from typing import Optional

import numpy as np


class ObjectInfo:
    """Identity (id) plus the category / score votes collected from detections and a counter of
    consecutive missed detections.  Hash/equality are by id, so an ObjectInfo can key a dict."""

    def __init__(self, id: int, category_id: Optional[int] = None, isthing: Optional[bool] = None,
                 score: Optional[float] = None):
        self.id = id
        self.category_ids = [category_id]
        self.scores = [score]
        self.isthing = isthing
        self.poke_count = 0  # detections since this object was last seen

    def poke(self) -> None:
        self.poke_count += 1

    def unpoke(self) -> None:
        self.poke_count = 0

    def merge(self, other) -> None:
        self.category_ids.extend(other.category_ids)
        self.scores.extend(other.scores)

    def vote_category_id(self) -> Optional[int]:
        votes = [c for c in self.category_ids if c is not None]
        if not votes:
            return None
        # scipy.stats.mode semantics (object_info.py:38): most frequent, smallest value on ties
        best = max(Counter(votes).items(), key=lambda kv: (kv[1], -kv[0]))
        return int(best[0])

    def vote_score(self) -> Optional[float]:
        scores = [s for s in self.scores if s is not None]
        return float(np.mean(scores)) if scores else None

    def get_rgb(self) -> np.ndarray:
        # panoptic-style id (0..255**3) -> RGB, little-endian base 256 (utils/pano_utils.py:7-15)
        return np.array([(self.id // 256**i) % 256 for i in range(3)], dtype=np.uint8)

    def copy_meta_info(self, other) -> None:
        self.category_ids = other.category_ids
        self.scores = other.scores
        self.isthing = other.isthing

    def __hash__(self):
        return hash(self.id)

    def __eq__(self, other):
        return self.id == other.id

    def __repr__(self):
        return f'(ID: {self.id}, cat: {self.category_ids}, isthing: {self.isthing}, score: {self.scores})'
