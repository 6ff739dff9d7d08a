"""gecco_amd -- MI355X-native linear-chain CRF inference for GECCO's ``gecco.crf`` hot path.

Scope: ``ClusterCRF.trained`` / ``predict_probabilities`` and the cluster
calling right behind it, behind GECCO's own ``crf_type=`` injection point.  Everything
numeric runs in hand-written HIP kernels for gfx950 through the C ABI in
``include/gecco_crf.h``; there is no CPU fallback.
"""
__version__ = "0.1.0"
