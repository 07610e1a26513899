"""Defaults of the ``full_gp`` path, value-for-value from the reference's ``src/config.py``
(:39-43 hyper-parameters, :53-56 ranges, :59-68 optimiser switches, :71-74 optimiser settings)."""

import math

import torch

NOISE_VARIANCE = (2.33e-6,)
OUTPUTSCALE_WIENER = 4.23e-13
OUTPUTSCALE_RBF = 0.0099
LENGTHSCALE_RBF = (12.11, 33.75, 45.14)

NOISE_VARIANCE_RANGE = (0, 1e5)
OUTPUTSCALE_WIENER_RANGE = (1e-15, 1e4)
OUTPUTSCALE_RBF_RANGE = (1e-10, 1e6)
LENGTHSCALE_RBF_RANGE = ((1e-5, 1e4), (1e-5, 1e4), (1e-5, 1e4))

HYPER_OPT_PARAMS = {
    "optimize": False,
    "parallelize": False,
    "opt_algorithm": "torch_adam",  # "torch_lbfgs", "botorch_lbfgs_B", "torch_adam"
}

OPTIM_MAX_ITER = 500
OPTIM_REL_TOL = 1e-5
OPTIM_LR = 1
DTYPE = torch.float64

INF = math.inf
