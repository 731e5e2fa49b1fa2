"""MI355X-native drop-in for the temporal-propagation path of `deva`
(hkchengrex/Tracking-Anything-with-DEVA).

This package provides the hot-path modules (`deva.model.network`, `deva.inference.inference_core`,
`memory_manager`, `kv_memory_store`, `object_manager`, `object_info`, `image_feature_store`,
`eval_args`, `deva.utils.tensor_utils`) on hand-written gfx950 kernels.  Everything else of the
reference tree (dataset readers, result savers, SAM / GroundingDINO adapters, metrics, ...) is
picked up unchanged from a reference checkout when one is on `sys.path` AFTER this package: every
`__init__` here extends its `__path__` over same-named packages further down `sys.path`, so
`evaluation/eval_vos.py` and `evaluation/eval_with_detections.py` run against this implementation
without edits.
"""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)

from deva.inference.inference_core import DEVAInferenceCore  # noqa: E402
from deva.model.network import DEVA  # noqa: E402
