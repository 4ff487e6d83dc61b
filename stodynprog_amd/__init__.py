"""stodynprog_amd -- MI355X-native stochastic dynamic programming.

Drop-in for `stodynprog` (pierre-haessig/stodynprog): same `SysDescription` /
`DPSolver` API (reference stodynprog/__init__.py:14), with the value-iteration
sweep, policy evaluation and multilinear interpolation running as hand-written
HIP kernels on gfx950 through the C ABI of include/sdp_hip.h.

    from stodynprog_amd import SysDescription, DPSolver
"""
from .sysdesc import SysDescription
from .solver import DPSolver
from .interp import MlinInterpolator

__version__ = '0.1.0'
__all__ = ['SysDescription', 'DPSolver', 'MlinInterpolator']
