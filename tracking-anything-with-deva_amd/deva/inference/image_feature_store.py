"""Per-frame cache of everything the key encoder produces (interface of
deva/inference/image_feature_store.py:7-48).

Several consumers of one frame -- the propagation step, the spatial alignments of a voting window,
the external processors -- ask for its multi-scale features and its key / shrinkage / selection;
the encoder runs on the first request only.  Entries live until `delete(index)`; a store that dies
non-empty warns, like the reference, because every entry pins ~100 MB of activations at 1080p."""
import warnings
from typing import Dict, Iterable, NamedTuple, Tuple

import torch


class _FrameFeatures(NamedTuple):
    ms_features: Iterable[torch.Tensor]  # f16, f8, f4
    pix_feat: torch.Tensor
    key: torch.Tensor
    shrinkage: torch.Tensor
    selection: torch.Tensor


class ImageFeatureStore:
    def __init__(self, network, no_warning: bool = False):
        self.network = network
        self.no_warning = no_warning
        self._store: Dict[int, _FrameFeatures] = {}
        self._pending: Dict[int, 'torch.cuda.Event'] = {}  # prefetch(): entries still being computed on the side stream
        self._side = None

    def _encode_feature(self, index: int, image: torch.Tensor) -> None:
        multi_scale, pix_feat = self.network.encode_image(image)
        self._store[index] = _FrameFeatures(multi_scale, pix_feat, *self.network.transform_key(pix_feat))

    def prefetch(self, index: int, image: torch.Tensor) -> None:
        """Extension (not in the reference): start the key encoder of frame `index` (1*3*H*W, padded like
        `DEVAInferenceCore` pads it) on a SIDE stream, so that it overlaps with the decoder / memory work of the frame
        the caller is about to step.  The key encoder depends on the image only; its batch-1 layers are launch-latency
        bound and leave most of the GPU idle, which the large decoder kernels of the previous frame fill.  The first
        `get_*` of that index waits for it.  Results are bit-identical to the unprefetched call (same kernels)."""
        if index in self._store or not image.is_cuda:
            return
        main = torch.cuda.current_stream(image.device)
        if self._side is None:
            self._side = torch.cuda.Stream(device=image.device)
        self._side.wait_stream(main)  # the image (and the weights) are ready
        with torch.cuda.stream(self._side):
            self._encode_feature(index, image)
            done = torch.cuda.Event()
            done.record(self._side)
        for t in (image, *self._tensors(self._store[index])):
            t.record_stream(self._side if t is image else main)  # allocator: produced on one stream, consumed on the other
        self._pending[index] = done

    @staticmethod
    def _tensors(entry: _FrameFeatures):
        return (*entry.ms_features, entry.pix_feat, entry.key, entry.shrinkage, entry.selection)

    def _entry(self, index: int, image: torch.Tensor) -> _FrameFeatures:
        done = self._pending.pop(index, None)
        if done is not None:
            torch.cuda.current_stream().wait_event(done)
        try:
            return self._store[index]
        except KeyError:
            self._encode_feature(index, image)
            return self._store[index]

    def get_ms_features(self, index, image) -> Iterable[torch.Tensor]:
        return self._entry(index, image).ms_features

    def get_key(self, index, image) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        e = self._entry(index, image)
        return e.key, e.shrinkage, e.selection

    def delete(self, index) -> None:
        self._pending.pop(index, None)
        self._store.pop(index, None)

    def __len__(self):
        return len(self._store)

    def __del__(self):
        if self._store and not self.no_warning:
            warnings.warn(f'Leaking {self._store.keys()} in the image feature store')
