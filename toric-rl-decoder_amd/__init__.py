"""toric-rl-decoder_amd: MI355X-native batched toric-code RL environment.

The env hot path of Lindeby/toric-RL-decoder (EnvSet / gym_ToricCode step loop,
generatePerspective, transitions, epsilon-greedy selection) as hand-written HIP kernels
for gfx950 behind a C-ABI (include/toricenv.h), with the reference's Python surface on top.
Import it as ``toric_rl_decoder_amd`` (the directory name has a hyphen).
"""
from ._lib import ToricEnvError, build, load, LIB_PATH  # noqa: F401
from .envset import (EnvSet, ToricEnv, TransitionBlock, alloc_stack, alloc_chunked, generatePerspectiveBatch,  # noqa: F401
                     generateTransitionParallel, make, to_structured, transition_dtype, SUPPORTED_SIZES,
                     configured_xcd_bias, set_xcd_bias)

from .policy import (NN_11, evaluate, predictMaxOptimized, seed_select, segment_max, selectActionBatch,  # noqa: F401,E402
                     selectActionEnvSet, _selectActionBatch_prime, prediction_smart,
                     generateNPlusQRandomErrors, generateNRandomErrors, generateRandomError)

from .actor import ExploreLoop, computePrioritiesParallel, run_actor  # noqa: F401,E402

__all__ = ["ExploreLoop", "computePrioritiesParallel", "run_actor", "NN_11", "evaluate", "predictMaxOptimized", "segment_max", "selectActionBatch", "selectActionEnvSet", "seed_select", "prediction_smart", "generateNPlusQRandomErrors", "EnvSet", "ToricEnv", "TransitionBlock", "alloc_stack", "alloc_chunked", "generatePerspectiveBatch", "generateTransitionParallel", "make", "to_structured",
           "transition_dtype", "ToricEnvError", "build", "load", "LIB_PATH", "SUPPORTED_SIZES"]
