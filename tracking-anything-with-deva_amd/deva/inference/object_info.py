"""What is known about one tracked object besides its mask (interface of
deva/inference/object_info.py:7-62): the votes for its category and score gathered from every
detection that was merged into it, whether it is a "thing", and for how many detection rounds in a
row it has gone unseen.  Identity is the id alone -- equality and hashing ignore the votes, which is
what lets `ObjectManager` key its tables with these objects."""
from collections import Counter
from typing import List, Optional

import numpy as np


class ObjectInfo:
    __slots__ = ('id', 'category_ids', 'scores', 'isthing', 'poke_count')

    def __init__(self, id: int, category_id: Optional[int] = None, isthing: Optional[bool] = None,
                 score: Optional[float] = None):
        self.id = id
        self.isthing = isthing
        self.category_ids: List[Optional[int]] = [category_id]
        self.scores: List[Optional[float]] = [score]
        self.poke_count = 0

    # ---- missed-detection bookkeeping (inference_core.py:185-196)
    def poke(self) -> None:
        self.poke_count = self.poke_count + 1

    def unpoke(self) -> None:
        self.poke_count = 0

    # ---- votes
    def merge(self, other) -> None:
        """a detection that matched this object contributes its votes"""
        self.category_ids += other.category_ids
        self.scores += other.scores

    def copy_meta_info(self, other) -> None:
        self.category_ids, self.scores, self.isthing = other.category_ids, other.scores, other.isthing

    def vote_category_id(self) -> Optional[int]:
        """most frequent category; the smallest one among equally frequent (scipy.stats.mode, :38)"""
        tally = Counter(c for c in self.category_ids if c is not None)
        if not tally:
            return None
        top = max(tally.values())
        return int(min(c for c, n in tally.items() if n == top))

    def vote_score(self) -> Optional[float]:
        known = [s for s in self.scores if s is not None]
        return float(np.mean(known)) if known else None

    def get_rgb(self) -> np.ndarray:
        """panoptic id -> colour, base-256 digits with the least significant first (pano_utils.py:7-15)"""
        digits, rest = [], int(self.id)
        for _ in range(3):
            rest, d = divmod(rest, 256)
            digits.append(d)
        return np.array(digits, dtype=np.uint8)

    def __eq__(self, other):
        return self.id == other.id

    def __hash__(self):
        return hash(self.id)

    def __repr__(self):
        return f'(ID: {self.id}, cat: {self.category_ids}, isthing: {self.isthing}, score: {self.scores})'
