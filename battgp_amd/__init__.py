"""battgp_amd - MI355X-native exact-GP engine for BattGP's ``full_gp`` hot path.

Covariance fill, jittered blocked Cholesky (fp64 MFMA trailing updates), triangular
solves, log-marginal likelihood and posterior mean/variance run as hand-written HIP
kernels for gfx950 behind a C-ABI (``include/battgp.h``), wrapped by Python classes
that mirror the reference's model surface (``BatteryCellGP_Full``,
``build_cellmodel_full``, ``ScaledRBFModel``).
"""

__version__ = "0.1.0"

KERNEL_BATTGP = 0
KERNEL_SCALED_RBF = 1
KERNEL_MATERN32 = 2
KERNEL_ARD_RBF = 3


def __getattr__(name):  # lazy: keep `import battgp_amd` cheap and torch-free
    if name in ("ExactGPEngine", "NotPSDError", "NumericalWarning", "EngineError"):
        from . import engine

        return getattr(engine, name)
    if name in ("BatteryCellGP_Full", "build_cellmodel_full"):
        from . import battcellgp_full

        return getattr(battcellgp_full, name)
    if name == "BatteryCellGP":
        from . import cell_gp

        return cell_gp.BatteryCellGP
    if name == "ScaledRBFModel":
        from . import standard_models

        return standard_models.ScaledRBFModel
    raise AttributeError(name)
