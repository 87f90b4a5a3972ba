"""Drop-in counterparts of the reference's ``modules`` package: same class and function
names, constructor/forward signatures and state-dict keys; the contrastive hot path inside
them runs on the HIP kernels."""
from . import models, seed, training_utils, training_utils_filip  # noqa: F401
