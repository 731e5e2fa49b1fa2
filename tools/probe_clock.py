#!/usr/bin/env python
"""bench.py's sustained fp32-MFMA probe on its own (run under rocprofv3 --pmc GRBM_GUI_ACTIVE: tools/evidence_r05.sh pmc)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402

print(bench.mfma_probe(torch.device('cuda:0')))
