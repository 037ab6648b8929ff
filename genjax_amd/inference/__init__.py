"""Inference library (reference: src/genjax/_src/inference/)."""
from .smc import ChangeTarget, Importance, ImportanceK, ParticleCollection, SMCAlgorithm, Target, TrialCollections  # noqa: F401
from .requests import HMC, IndexRequest, Regenerate, Rejuvenate, SafeHMC, StaticRequest, Update  # noqa: F401
from .pf import BootstrapFilter, LinearGaussianSSM, resample  # noqa: F401
