"""genjax_amd — MI355X-native inference-kernel backend with the GenJAX API shape.

Host-side Python keeps the reference's surface (``gen``, ``Target``, ``ChoiceMap``/``C``,
``Selection``/``S``, ``ImportanceK``, ``HMC`` ...); everything per-particle runs in hand-written HIP
kernels behind the C ABI of ``include/gjx.h`` (``genjax_amd/csrc/libgjx_hip.so``).  There is no CPU
fallback: compute entry points raise ``GjxError`` if the library is missing.
"""
from . import config, inference  # noqa: F401
from .core import (C, ChoiceMap, ChoiceMapBuilder, ChoiceMapNoValueAtAddress, Diff, Mask, NoChange, S, Selection, SelectionBuilder,  # noqa: F401
                   UnknownChange, fold_in, key, split)
from .gen import (Distribution, Marginal, array, chi2, dirichlet, geometric, gumbel, half_cauchy, inverse_gamma,
                  logit_normal, marginal, poisson, student_t, truncated_normal, weibull, StaticGenerativeFunction, Trace, bernoulli, beta, categorical,  # noqa: F401
                  cauchy, cond, const, exp, exponential, flip, gamma, gen, half_normal, laplace, log_normal,
                  iterate, iterate_final, mv_normal_diag, normal, repeat, scan, sigmoid, softplus, take, uniform, vmap, where)
from .gen import Scan, Vmap, accumulate, reduce  # noqa: F401
from .gen import (chi, double_sided_maxwell, exp_gamma, exp_inverse_gamma, half_student_t, inverse_gaussian, kumaraswamy, moyal,  # noqa: F401
                  negative_binomial, truncated_cauchy, von_mises)
from .gen import Expr, NotSupportedInModelBody, cos, dot, log, log1p, maximum, minimum, sin, sqrt, square, tanh  # noqa: F401  (general expressions: GJX_P_EXPR)
from .inference import requests, smc  # noqa: F401  (the reference's genjax.smc / genjax.requests modules)
from .inference.smc import SMCAlgorithm as Algorithm  # noqa: F401
from .inference import (HMC, IndexRequest, BootstrapFilter, ChangeTarget, Importance, ImportanceK, LinearGaussianSSM,  # noqa: F401
                        ParticleCollection, Regenerate, Rejuvenate, SafeHMC, SMCAlgorithm, StaticRequest, Target, TrialCollections, Update)
from .program import AddressReuse, MissingAddress  # noqa: F401

__version__ = "0.1.0"
