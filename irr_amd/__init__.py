"""irr_amd -- MI355X-native IRR-PWC forward/backward path (drop-in for visinf/irr's models/IRR_PWC.py).

Only what the hot path needs lives here: ``csrc/`` (HIP kernels + the C ABI of include/irr_hip.h),
``hip.py`` (ctypes binding), ``functional.py`` (autograd operators) and the host-side mirror of the
reference's module surface.
"""
__version__ = "0.1.0"
