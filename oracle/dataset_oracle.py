"""oracle/dataset_oracle.py — CPU restatement of the VO dataset's per-sample processing (TEST INFRASTRUCTURE ONLY).

Only tests/ may import this module; the product path (pointnav-vo_amd/dataset.py) never does.

Restates, in plain numpy, one sample at a time:
  * StatePairRegressionDataset._process_data   /root/reference/pointnav_vo/vo/dataset/regression_geo_invariance_iter_dataset.py:205-454
  * BaseRegressionDataset._discretize_depth_func  .../vo/dataset/regression_iter_dataset.py:32-69
  * NormalizedDepth2TopDownViewHabitat (the numpy twin)  /root/reference/pointnav_vo/utils/geometry_utils.py:275-470

Pinned by tests/golden/dataset_*.npz, captured by tests/golden/gen_golden_dataset.py from the reference's own
`_process_data` run on seeded chunks.  PARITY UNPINNED, as everywhere in this repo, for two third-party steps whose code is
not under /root/reference: cv2.GaussianBlur (OpenCV; restated as the ksize-3 {1/4,1/2,1/4} separable kernel with a zero
border) and the quaternion algebra behind the swapped entries' targets (habitat-lab `agent_state_target2ref`,
`quaternion_from_coeff`, `quaternion_rotate_vector` and the numpy-quaternion package; restated from their published
definitions).  The golden generator uses the same restatements as stand-ins for the absent packages.
"""
import numpy as np

CUR_REL_TO_PREV, PREV_REL_TO_CUR = 0, 1
MOVE_FORWARD, TURN_LEFT, TURN_RIGHT = 1, 2, 3


def discretize_depth(raw_depth, bins):
    """regression_iter_dataset.py:32-69: one-hot uint8 [H,W,bins]; edges i/bins are python floats, which numpy casts to the
    array's dtype (float16 for HDF5 depth) before comparing."""
    dt = raw_depth.dtype
    edges = [np.asarray(i * 1.0 / bins).astype(dt) for i in range(bins)] + [np.asarray(1.0).astype(dt)]
    out = np.zeros(raw_depth.shape + (bins,), dtype=np.uint8)
    for i in range(bins):
        hi = (raw_depth <= edges[i + 1]) if i == bins - 1 else (raw_depth < edges[i + 1])
        out[..., i] = (raw_depth >= edges[i]) & hi
    return out


def blur3(src):
    """cv2.GaussianBlur(src, (3,3), 0, 0, BORDER_ISOLATED) on float32 (OpenCV's fixed kernel for ksize 3: 1/4, 1/2, 1/4)."""
    s = np.ascontiguousarray(src, dtype=np.float32)
    q, h = np.float32(0.25), np.float32(0.5)
    p = np.pad(s, ((0, 0), (1, 1)))
    t = (s * h + (p[:, :-2] + p[:, 2:]) * q).astype(np.float32)
    p = np.pad(t, ((1, 1), (0, 0)))
    return (t * h + (p[:-2, :] + p[2:, :]) * q).astype(np.float32)


def top_down_view(depth, min_depth, max_depth, vis_size_h, vis_size_w, hfov_rad, rows_around_center=50, eps=0.01):
    """geometry_utils.py:293-470 (flag_center_crop=True, ksize=3).  depth [H,W] in [0,1] -> float64 [H,W]."""
    H, W = vis_size_h, vis_size_w
    rows = np.nonzero(depth.astype(np.float64).sum(axis=1) > 0)[0]
    cols = np.nonzero(depth.astype(np.float64).sum(axis=0) > 0)[0]
    if rows.size == 0:
        return np.zeros((H, W))
    r0, r1, c0, c1 = rows[0], rows[-1], cols[0], cols[-1]
    crop = blur3(depth[r0:r1 + 1, c0:c1 + 1].astype(np.float32))
    hc = crop.shape[0]
    half = int(np.ceil(hc / 2))
    b0, b1 = max(0, half - rows_around_center), min(hc, half + rows_around_center)
    f = (W / 2) / np.tan(hfov_rad / 2)
    kinv = np.linalg.inv(np.array([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1.0]]))
    u = np.arange(crop.shape[1], dtype=np.float64) + c0 + 0.5
    xc = kinv[0, 0] * u + kinv[0, 2]
    z = (crop[b0:b1, :] * np.float32(max_depth - min_depth) + np.float32(min_depth)).astype(np.float64)   # float32 true depth
    X = xc[None, :] * z
    right = (kinv @ np.array([W - 0.5, 0, 1.0]) * max_depth)[0]
    min_x, x_range = -right, 2 * right
    xn = (X - min_x) / (x_range * (1 + eps))
    zn = (z - min_depth) / ((max_depth - min_depth) * (1 + eps))
    row = (H - np.ceil(H * zn)).astype(np.int64)
    col = np.floor(W * xn).astype(np.int64)
    ok = (row >= 0) & (row < H) & (col >= 0) & (col < W)
    cnt = np.zeros((H, W))
    np.add.at(cnt, (row[ok], col[ok]), 1)
    if cnt.max() == 0:
        return cnt
    return np.minimum(cnt / cnt.max(), 1.0)


def quat_mul(a, b):                       # [x, y, z, w]
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz])


def state_target2ref(ref_rot, ref_pos, tgt_rot, tgt_pos):
    """habitat-lab agent_state_target2ref: (ref^-1 * target, ref^-1 (target_pos - ref_pos) ref), rotations normalised."""
    r = np.asarray(ref_rot, dtype=np.float64)
    t = np.asarray(tgt_rot, dtype=np.float64)
    r, t = r / np.sqrt((r * r).sum()), t / np.sqrt((t * t).sum())
    rinv = r * np.array([-1.0, -1.0, -1.0, 1.0])
    d = (np.asarray(tgt_pos) - np.asarray(ref_pos)).astype(np.float64)
    v = quat_mul(quat_mul(rinv, np.array([d[0], d[1], d[2], 0.0])), r)
    return quat_mul(rinv, t), v[:3]


def process_sample(chunk, i, *, H, W, act_type=-1, bins=0, tdv_infos=None, geo=()):
    """One call of _process_data: list of entries (dict) in emission order."""
    a = int(chunk["actions"][i])
    fr = {}
    for k in ("prev", "cur"):
        rgb = chunk[f"{k}_rgbs"][i].reshape(H, W, 3)
        d = chunk[f"{k}_depths"][i].reshape(H, W, 1)
        dd = discretize_depth(d[..., 0], bins) if bins else np.zeros((H, W, 1))
        tdv = top_down_view(d[..., 0], **tdv_infos)[..., None] if tdv_infos else np.zeros((H, W, 1))
        fr[k] = (rgb, d, dd, tdv)
    out = []

    def entry(first, second, action, dtype, tgt):
        out.append(dict(action=action, data_type=dtype,
                        rgb=np.concatenate([fr[first][0], fr[second][0]], 2).astype(np.float32),
                        depth=np.concatenate([fr[first][1], fr[second][1]], 2).astype(np.float32),
                        dd=np.concatenate([fr[first][2], fr[second][2]], 2).astype(np.float32),
                        tdv=np.concatenate([fr[first][3], fr[second][3]], 2).astype(np.float32),
                        target=np.asarray(tgt, dtype=np.float32)))

    if act_type == -1 or (isinstance(act_type, int) and a == act_type) or "inverse_joint_train" in geo:
        dp, dr = chunk["delta_positions"][i], chunk["delta_rotations"][i]
        entry("prev", "cur", a, CUR_REL_TO_PREV, [dp[0], dp[1], dp[2], 2 * np.arctan2(dr[1], dr[3])])
    flag1 = act_type != -1 and "inverse_data_augment_only" in geo and a != MOVE_FORWARD and a != act_type
    flag2 = act_type != -1 and a != MOVE_FORWARD and "inverse_joint_train" in geo
    if flag1 or flag2:
        q, p = state_target2ref(chunk["cur_global_rotations"][i], chunk["cur_global_positions"][i],
                                chunk["prev_global_rotations"][i], chunk["prev_global_positions"][i])
        q = q.astype(np.float32)
        entry("cur", "prev", TURN_LEFT if a == TURN_RIGHT else TURN_RIGHT, PREV_REL_TO_CUR,
              [p[0], p[1], p[2], 2 * np.arctan2(q[1], q[3])])
    return out
