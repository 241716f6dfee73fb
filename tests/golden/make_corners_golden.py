#!/usr/bin/env python3
"""Generates tests/golden/corners_small.json: a 260 x 200 window of a rendered Triple Sphere chessboard view and the
corner candidates / board the CPU oracle finds in it (oracle/tscm_oracle_corners.c, tscm_oracle_boards.c).
A regression fixture for the oracle and a fixed target for the GPU path; it does NOT pin parity with OpenCV (there is
no OpenCV in this container).  Run from the repository root."""
import base64
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import pyoracle as orc  # noqa: E402
from tscm_calib_amd import synth  # noqa: E402

p = synth.make_problem(1, 6, 3, noise_px=0.0, perturb=False)
full = synth.render_chessboard(p.meta["gt_intr"][0], p.meta["gt_board_rt"][0], 9, 6, 45.0, 1280, 1080, supersample=3)
o = p.view_offset[0]
uv = np.stack([p.obs_u[o:o + 54], p.obs_v[o:o + 54]], axis=1)
x0, y0 = int(uv[:, 0].min()) - 30, int(uv[:, 1].min()) - 30
img = np.ascontiguousarray(full[y0:y0 + 200, x0:x0 + 260])
d = orc.detect_corners(img)
keep = d["score"] >= 0.01
boards = orc.chessboards_from_corners(d["x"][keep], d["y"][keep], d["v1"][keep], d["v2"][keep])
out = dict(width=260, height=200, origin=[x0, y0], image_b64=base64.b64encode(img.tobytes()).decode(),
           n_maxima=int(d["n"]), x=d["x"].tolist(), y=d["y"].tolist(), v1=d["v1"].tolist(), v2=d["v2"].tolist(),
           score=d["score"].tolist(), sub=d["sub"].tolist(), boards=[b.tolist() for b in boards],
           truth=(uv - [x0, y0]).tolist())
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "corners_small.json"), "w") as f:
    json.dump(out, f)
print("maxima", d["n"], "kept", int(keep.sum()), "boards", [b.shape for b in boards])
