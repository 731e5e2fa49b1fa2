"""TEST INFRASTRUCTURE: run one of the reference's UNCHANGED evaluation drivers (evaluation/eval_vos.py,
evaluation/eval_with_detections.py) in this process, with

* sys.path = [overlay package (optional), reference checkout]  -- INTEGRATION.md's PYTHONPATH order;
* test-only stand-ins for the third-party modules the readers / savers import and this image lacks
  (torchvision.transforms, pycocotools.mask, supervision, hickle) -- none of them is on the hot path;
* without a GPU: the HIP ops replaced by their PyTorch emulation (tests/emu_ops.py) and the `.cuda()` /
  `torch.cuda.Event` calls of the drivers turned into no-ops;
* with `--reference-only`: no overlay at all (the reference's own PyTorch path on the CPU) and its PuLP
  solver replaced by the exact enumeration, so that the same driver gives the expected outputs.

usage: python tests/run_reference_driver.py [--reference-only] <driver.py> [driver args...]
"""
import enum
import os
import runpy
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get('DEVA_REFERENCE_ROOT', '/root/reference')


def install_third_party_stubs():
    from PIL import Image

    tv = types.ModuleType('torchvision')
    tr = types.ModuleType('torchvision.transforms')

    class InterpolationMode(enum.Enum):
        NEAREST = 'nearest'
        BILINEAR = 'bilinear'

    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    class ToTensor:
        def __call__(self, img):
            return torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float() / 255

    class Normalize:
        def __init__(self, mean, std):
            self.mean, self.std = torch.tensor(mean).view(-1, 1, 1), torch.tensor(std).view(-1, 1, 1)

        def __call__(self, x):
            return (x - self.mean) / self.std

    def _out_size(h, w, size):
        short, long = (w, h) if w <= h else (h, w)
        if short == size:
            return h, w
        new_short, new_long = size, int(size * long / short)
        return (new_long, new_short) if w <= h else (new_short, new_long)

    class Resize:
        def __init__(self, size, interpolation=InterpolationMode.BILINEAR, antialias=None):
            self.size, self.interpolation, self.antialias = size, interpolation, antialias

        def __call__(self, x):
            if isinstance(x, Image.Image):
                oh, ow = _out_size(x.height, x.width, self.size)
                if (oh, ow) == (x.height, x.width):
                    return x
                mode = Image.NEAREST if self.interpolation == InterpolationMode.NEAREST else Image.BILINEAR
                return x.resize((ow, oh), mode)
            oh, ow = _out_size(x.shape[-2], x.shape[-1], self.size)
            if (oh, ow) == tuple(x.shape[-2:]):
                return x
            if self.interpolation == InterpolationMode.NEAREST:
                return F.interpolate(x.unsqueeze(0), (oh, ow), mode='nearest')[0]
            return F.interpolate(x.unsqueeze(0), (oh, ow), mode='bilinear', align_corners=False,
                                 antialias=bool(self.antialias))[0]

    tr.InterpolationMode, tr.Compose, tr.ToTensor, tr.Normalize, tr.Resize = InterpolationMode, Compose, ToTensor, Normalize, Resize
    tv.transforms = tr
    tv.ops = types.ModuleType('torchvision.ops')
    sys.modules.update({'torchvision': tv, 'torchvision.transforms': tr, 'torchvision.ops': tv.ops})
    coco = types.ModuleType('pycocotools')
    coco.mask = types.ModuleType('pycocotools.mask')
    sys.modules.update({'pycocotools': coco, 'pycocotools.mask': coco.mask})
    sys.modules.setdefault('supervision', types.ModuleType('supervision'))


def install_cpu_shims():
    """the drivers call .cuda() and time with CUDA events; on a box without a GPU those become no-ops"""
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self

    class _Event:
        def __init__(self, enable_timing=False):
            import time
            self._t, self._time = 0.0, time

        def record(self):
            self._t = self._time.perf_counter()

        def elapsed_time(self, other):
            return max((other._t - self._t) * 1e3, 1e-3)

    torch.cuda.Event = _Event
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.max_memory_allocated = lambda *a, **k: 0


def main():
    argv = sys.argv[1:]
    reference_only = argv and argv[0] == '--reference-only'
    if reference_only:
        argv = argv[1:]
    driver, sys.argv = argv[0], argv
    paths = [REF] if reference_only else [os.path.join(ROOT, 'tracking-anything-with-deva_amd'), REF]
    sys.path[:0] = paths + [HERE, ROOT]
    install_third_party_stubs()
    torch.manual_seed(0)
    np.random.seed(0)
    if not torch.cuda.is_available():
        install_cpu_shims()
    if reference_only:
        sys.modules.setdefault('pulp', types.ModuleType('pulp'))
        import deva.model.resnet as R  # no network: the ImageNet initialisation is overwritten by --model anyway
        r18, r50 = R.resnet18, R.resnet50
        R.resnet18 = lambda pretrained=True, extra_dim=0: r18(pretrained=False, extra_dim=extra_dim)
        R.resnet50 = lambda pretrained=True, extra_dim=0: r50(pretrained=False, extra_dim=extra_dim)
        import deva.inference.consensus_automatic as CA
        src = open(os.path.join(ROOT, 'tracking-anything-with-deva_amd', 'deva', 'inference', 'consensus_automatic.py')).read()
        ns = {}
        exec(src[src.index('def solve_exact'):src.index('def solve(')], {'np': np, 'List': list, 'Tuple': tuple}, ns)
        CA.solve_with_pulp, CA.use_gurobi = ns['solve_exact'], False
    elif not torch.cuda.is_available():
        import emu_ops

        class _Setter:
            @staticmethod
            def setattr(obj, name, value):
                setattr(obj, name, value)

        emu_ops.install(_Setter)
    torch.set_num_threads(max(1, (os.cpu_count() or 2) // 2))
    runpy.run_path(driver, run_name='__main__')


if __name__ == '__main__':
    main()
