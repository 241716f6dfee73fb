"""tscm_calib_amd -- MI355X-native Levenberg-Marquardt solver for the Triple Sphere camera model.

Only what the reprojection-error LM hot path of imuncle/TSCM_Calib and its immediate neighbours need:
  csrc/        HIP kernels (gfx950) + host LM driver + the C ABI (include/tscm/tscm.h)
  lib.py       ctypes binding of the C ABI
  api.py       host-side mirror of the reference interface (calibrate / refinement / functor eval)
  rig.py       rig initialisation (MultiCalib constructor), focal estimate, [r1 r2 t] -> pose
  maps.py      remap tables (undistort, undistort_chessboard, epipolar rectification)
  calib_io.py  calibration YAML in the cv::FileStorage layout
  problem.py   problem container, frame sharding
  synth.py     deterministic synthetic chessboard observations (BASELINE.json configs)
"""
from .problem import Problem, shard_frames  # noqa: F401
