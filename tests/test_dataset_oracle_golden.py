"""CPU: oracle/dataset_oracle.py (numpy restatement of the dataset's per-sample processing) against golden vectors captured
from the reference's own StatePairRegressionDataset._process_data (tests/golden/gen_golden_dataset.py)."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import dataset_oracle as do
from pointnav_vo_amd import synth
from pointnav_vo_amd.dataset import entries_of_chunk

CASES = ["dataset_64x48_joint.npz", "dataset_70x40_all.npz", "dataset_341x192_left_aug.npz", "dataset_64x48_rgbd_fwd.npz"]


def case(rec):
    N, W, H, seed, bins = (int(rec[k]) for k in ("N", "W", "H", "seed", "bins"))
    at = rec["act_type"]
    act_type = int(at) if at.ndim == 0 else [int(x) for x in at]
    geo = tuple(g for g in str(rec["geo"]).split(",") if g)
    infos = dict(min_depth=0.1, max_depth=10.0, vis_size_h=H, vis_size_w=W, hfov_rad=np.deg2rad(70.0),
                 rows_around_center=int(rec["rows_around_center"])) if int(rec["has_tdv"]) else None
    return synth.make_dataset_chunk(N, H, W, seed=seed, bins=max(bins, 1)), dict(H=H, W=W, act_type=act_type, bins=bins,
                                                                               tdv_infos=infos, geo=geo)


@pytest.mark.parametrize("fname", CASES)
def test_oracle_matches_reference_process_data(fname):
    rec = load_golden(fname)
    ch, kw = case(rec)
    ent = entries_of_chunk(ch["actions"], kw["act_type"], kw["geo"])
    samples = sorted({e[0] for e in ent}, key=[e[0] for e in ent].index)
    out = [e for i in samples for e in do.process_sample(ch, i, **kw)]
    assert len(out) == rec["actions"].shape[0] == len(ent)
    # the host-side entry enumeration of the product == the oracle's == the reference's
    assert [e[2] for e in ent] == [e["action"] for e in out] == rec["actions"].reshape(-1).tolist()
    assert [e[3] for e in ent] == [e["data_type"] for e in out] == rec["data_types"].reshape(-1).astype(int).tolist()
    assert [e[0] for e in ent] == rec["entry_idxs"].reshape(-1).astype(int).tolist()
    for m, e in enumerate(out):
        np.testing.assert_array_equal(e["target"], rec["targets"][m])                       # float16 poses: exact
        np.testing.assert_array_equal(e["rgb"].astype(np.float64).sum((0, 1)), rec["rgb_sum"][m])
        np.testing.assert_array_equal(e["depth"].astype(np.float64).sum((0, 1)), rec["depth_sum"][m])
        if kw["bins"]:
            b = kw["bins"]
            assert (e["dd"].reshape(kw["H"], kw["W"], 2, b).sum(-1) == 1).all()
            np.testing.assert_array_equal(e["dd"].reshape(kw["H"], kw["W"], 2, b).argmax(-1), rec["dd_bin"][m])
        np.testing.assert_array_equal(e["tdv"], rec["tdv_pairs"][m])                         # integer histogram / max: exact


def test_float16_edges_differ_from_float32_edges():
    """Why the dataset path needs its own edges: float16(0.1) < 0.1, so the float16 value 0.0999755859375 lands in bin 1
    under the dataset's comparison and in bin 0 under the nav-time float32 one."""
    d = np.array([[np.float16(0.1)]], dtype=np.float16)
    assert do.discretize_depth(d, 10)[0, 0].argmax() == 1
    assert do.discretize_depth(d.astype(np.float32), 10)[0, 0].argmax() == 0
