"""Multilinear interpolation (same names as the reference's
stodynprog/dolointerpolation/__init__.py:3-4), backed by the HIP kernel."""
from ..interp import MultilinearInterpolator, multilinear_interpolation, mlinspace

__all__ = ['MultilinearInterpolator', 'multilinear_interpolation', 'mlinspace']
