"""Pins the one unpinned step of the ego top-down view — cv2.GaussianBlur(crop, (3, 3), sigmaX=0, sigmaY=0,
borderType=cv2.BORDER_ISOLATED) at /root/reference/pointnav_vo/utils/geometry_utils.py:528-535 — wherever an OpenCV is
importable ("verify on any box with cv2").  OpenCV is absent from this image, so these tests skip here;
the oracle's and the HIP kernel's blur are restated from OpenCV's published algorithm ({1/4, 1/2, 1/4} separable, symmetric
taps added first, constant-0 border).  Bit-for-bit on every top-down case of tests/golden/preproc.npz."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import oracle

cv2 = pytest.importorskip("cv2")

CASES = ["full_uniform", "full_border", "full_near", "full_fp32", "full_one_pixel", "full_top_band", "small_odd", "small_border"]


def _cv2_blur(crop):
    return cv2.GaussianBlur(np.ascontiguousarray(crop, dtype=np.float32), (3, 3), sigmaX=0, sigmaY=0, borderType=cv2.BORDER_ISOLATED)


def _crop(d, bbox):
    return d[bbox[0]:bbox[1] + 1, bbox[2]:bbox[3] + 1]


@pytest.mark.parametrize("case", CASES)
def test_oracle_blur_equals_cv2(case):
    rec = load_golden("preproc.npz")
    d = rec[f"tdv_in/{case}"].astype(np.float32).reshape(rec[f"tdv_in/{case}"].shape[:2])
    H, W = d.shape
    c = oracle.topdown_consts(H, W, 70, 0.1, 10.0)
    out, aux = oracle.topdown_view(d, c, return_aux=True)
    assert not aux["empty"]
    want = _cv2_blur(_crop(d, aux["bbox"]))
    np.testing.assert_array_equal(aux["blur"], want)
    # and the whole view with cv2's blur substituted for the oracle's
    np.testing.assert_array_equal(oracle.topdown_view(d, c, blur_in=want), out)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_hip_topdown_view_equals_the_pipeline_with_cv2_blur(case):
    import torch
    from pointnav_vo_amd.trainer import NormalizedDepth2TopDownViewHabitatTorch
    rec = load_golden("preproc.npz")
    d = rec[f"tdv_in/{case}"].astype(np.float32).reshape(rec[f"tdv_in/{case}"].shape[:2])
    H, W = d.shape
    c = oracle.topdown_consts(H, W, 70, 0.1, 10.0)
    _, aux = oracle.topdown_view(d, c, return_aux=True)
    want = oracle.topdown_view(d, c, blur_in=_cv2_blur(_crop(d, aux["bbox"])))
    gen = NormalizedDepth2TopDownViewHabitatTorch(min_depth=0.1, max_depth=10.0, vis_size_h=H, vis_size_w=W, hfov_rad=70)
    got = gen.gen_top_down_view(torch.from_numpy(d[..., None]).to("cuda:0")).cpu().numpy()
    np.testing.assert_array_equal(got, want)
