"""irr_amd -- MI355X-native IRR-PWC forward/backward path (drop-in for visinf/irr's models/IRR_PWC.py).

Only what the hot path needs lives here: ``csrc/`` (HIP kernels + the C ABI of include/irr_hip.h),
``hip.py`` (ctypes binding), ``functional.py`` / ``conv.py`` (autograd operators) and the host-side mirror
of the reference's module surface (``irr_pwc.PWCNet``, ``modules``, ``losses``, ``correlation.Correlation``).
"""
__version__ = "0.1.0"

from .irr_pwc import PWCNet  # noqa: E402,F401
from .correlation import Correlation  # noqa: E402,F401
from .functional import compute_cost_volume  # noqa: E402,F401
from .losses import MultiScaleEPE_PWC_Bi_Occ_upsample  # noqa: E402,F401

from . import ddp, optim, train  # noqa: E402,F401

from . import pwcnet as _pwcnet  # noqa: E402
from .pwcnet_variants import (PWCNet_bi, PWCNet_occ, PWCNet_occ_bi, PWCNet_irr, PWCNet_irr_bi,  # noqa: E402,F401
                              PWCNet_irr_occ, PWCNet_irr_occ_bi)                                # models/__init__.py:28-34
from .losses import (MultiScaleEPE_PWC, MultiScaleEPE_PWC_Bi, MultiScaleEPE_PWC_Occ, MultiScaleEPE_PWC_Bi_Occ,  # noqa: E402,F401
                     MultiScaleEPE_PWC_Bi_Occ_upsample_Sintel, MultiScaleEPE_PWC_Bi_Occ_upsample_KITTI)

PWCNet_baseline = _pwcnet.PWCNet       # models/__init__.py:27 `PWCNet = pwcnet.PWCNet` (ablation baseline, config 0)
IRR_PWC = PWCNet          # models/__init__.py:35 rebinds the module name to the class
