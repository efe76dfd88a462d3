"""Constants of the VO path; values as in /root/reference/pointnav_vo/vo/common/common_vars.py:9-57."""
import numpy as np

EVAL_BATCHSIZE = 64
N_ACTS = 4

UNIFIED = -1
STOP = 0
MOVE_FORWARD = 1
TURN_LEFT = 2
TURN_RIGHT = 3
ACT_IDX2NAME = {UNIFIED: "unified", MOVE_FORWARD: "forward", TURN_LEFT: "left", TURN_RIGHT: "right"}
ACT_NAME2IDX = {"forward": MOVE_FORWARD, "left": TURN_LEFT, "right": TURN_RIGHT, "all": -1}

# [x, z, w]
NO_NOISE_DELTAS = {
    MOVE_FORWARD: [0.0, -0.25, 0.0],
    TURN_LEFT: [0.0, 0.0, np.radians(10)],
    TURN_RIGHT: [0.0, 0.0, -np.radians(10)],
}
DEFAULT_DELTA_TYPES = ["dx", "dz", "dyaw"]

EMBED_DIM = 32
RGB_PAIR_CHANNEL = 6
DEPTH_PAIR_CHANNEL = 2
TOP_DOWN_VIEW_PAIR_CHANNEL = 2
DEFAULT_DELTA_STATE_SIZE = 4
