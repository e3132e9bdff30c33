"""vjf_amd -- MI355X-native implementation of VJF's online variational filtering step.

Same module layout and operator API as the reference package `vjf` (catniplab/vjf):
`vjf_amd.model.VJF`, `vjf_amd.module.RBF / LinearRegression`, `vjf_amd.likelihood.*`,
`vjf_amd.recognition.Recognition`, `vjf_amd.functional.*`, `vjf_amd.distribution.Gaussian`.
All arithmetic runs in hand-written HIP kernels (csrc/) behind a C ABI (include/vjf_hip.h);
there is no CPU compute path.
"""
import logging

from . import distribution, functional, likelihood, model, module, recognition, util  # noqa: F401
from .distribution import Gaussian  # noqa: F401
from .model import VJF  # noqa: F401

logging.basicConfig(level=logging.INFO, format="%(asctime)s  %(message)s")   # vjf/__init__.py:4

__version__ = "0.1.0"
