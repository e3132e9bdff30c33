"""Posterior / predictive carrier type (mirror of vjf/distribution.py:3)."""
from collections import namedtuple

Gaussian = namedtuple('Gaussian', ['mean', 'logvar'])
