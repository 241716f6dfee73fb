"""tscm_calib_amd -- MI355X-native Levenberg-Marquardt solver for the Triple Sphere camera model.

Only what the reprojection-error LM hot path of imuncle/TSCM_Calib needs:
  csrc/      HIP kernels (gfx950) + host LM driver + the C ABI (include/tscm/tscm.h)
  lib.py     ctypes binding of the C ABI
  api.py     host-side mirror of the reference interface (calibrate / refinement / functor eval)
  problem.py problem container, frame sharding
  synth.py   deterministic synthetic chessboard observations (BASELINE.json configs)
"""
from .problem import Problem, shard_frames  # noqa: F401
