"""Pin oracle/geometry.py against the reference's camera / resample code."""
import numpy as np

from conftest import load_npz
from oracle import geometry as og


def _cams(g, i):
    W, H = (int(v) for v in g[f"{i}/src_wh"])
    src = og.Fisheye62(W, H, tuple(g[f"{i}/src_f"]), tuple(g[f"{i}/src_c"]), list(g[f"{i}/src_coeffs"]),
                       g[f"{i}/src_c2w"])
    dst = og.Pinhole(96, 96, tuple(g[f"{i}/dst_f"]), tuple(g[f"{i}/dst_c"]), g[f"{i}/dst_c2w"])
    return src, dst


def test_warp_maps_match_reference_camera_classes():
    g = load_npz("warp_maps.npz")
    for i in range(int(g["n_cases"])):
        src, dst = _cams(g, i)
        mx, my = og.fisheye_warp_maps(src, dst)
        # identical fp64 formulas -> identical float32 maps (allow 1 ulp of float32 at ~600 px)
        np.testing.assert_allclose(mx, g[f"{i}/map_x"], rtol=0, atol=1e-4)
        np.testing.assert_allclose(my, g[f"{i}/map_y"], rtol=0, atol=1e-4)
        assert int((mx == -1).sum()) >= int(g[f"{i}/n_behind"])


def test_resample_pinhole_matches_reference():
    g = load_npz("resample.npz")
    out = og.resample_pinhole(g["images"], g["xfs"], (96, 96))
    assert 0.3 < (g["out"] != 0).mean() < 0.95        # fixture really has in- and out-of-image pixels
    np.testing.assert_allclose(out, g["out"], rtol=0, atol=1e-3)   # grey levels 0..255
    for v in range(g["images"].shape[0]):
        xf = og.resample_matrix(g["orig_K"][v], g["orig_ext"][v], g["new_K"][v].astype(np.float64),
                                np.linalg.inv(g["new_E"][v].astype(np.float64)))
        np.testing.assert_allclose(xf, g["xfs"][v], rtol=2e-4, atol=2e-3)


def test_remap_u8_properties():
    """cv2.remap is unpinned (no cv2 here): properties of the restated algorithm."""
    rng = np.random.default_rng(5)
    src = rng.integers(0, 256, size=(40, 50)).astype(np.uint8)
    yy, xx = np.mgrid[0:40, 0:50].astype(np.float32)
    np.testing.assert_array_equal(og.remap_linear_u8(src, xx, yy), src)              # identity map
    np.testing.assert_array_equal(og.remap_linear_u8(src, xx + 3, yy)[:, :47], src[:, 3:])
    assert (og.remap_linear_u8(src, xx + 3, yy)[:, 47:] == 0).all()                  # constant-0 border
    const = np.full((40, 50), 201, np.uint8)
    mx = rng.uniform(1, 48, size=(30, 30)).astype(np.float32)
    my = rng.uniform(1, 38, size=(30, 30)).astype(np.float32)
    np.testing.assert_array_equal(og.remap_linear_u8(const, mx, my), 201)
    a = og.remap_linear_u8(src, mx, my).astype(np.float64)
    b = og.remap_linear_f32(src, mx, my).astype(np.float64)
    # 1/32-px coordinate quantisation moves the sample by <= 1/64 px per axis
    assert np.abs(a - b).max() <= 0.5 + 255 * (2 / 64), np.abs(a - b).max()
    assert (og.remap_linear_u8(src, np.full((2, 2), -1, np.float32), np.full((2, 2), -1, np.float32)) == 0).all()
