"""CPU checks of the numpy Canny standing in for OpenCV in apps/canny2image.py (reference apps/gradio_canny2image.py:72-75
`apply_canny(img, low, high)`; SURVEY.md section 8 (f)4 "CPU-side data path").  OpenCV is not in this image, so the detector is
pinned by its stages: Sobel gradients against scipy.ndimage, known-answer contours, hysteresis and threshold monotonicity."""
import importlib.util
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("canny2image", os.path.join(ROOT, "apps", "canny2image.py"))
app = importlib.util.module_from_spec(spec)
spec.loader.exec_module(app)


def test_sobel_gradient_matches_scipy():
    ndi = pytest.importorskip("scipy.ndimage")
    rng = np.random.default_rng(0)
    g = rng.integers(0, 256, (37, 53)).astype(np.float32)
    kx = np.array([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]], np.float32)
    from controllora_amd.process import _conv2_same
    gx, gy = _conv2_same(g, kx), _conv2_same(g, kx.T)
    sx, sy = ndi.sobel(g, axis=1, mode="nearest"), ndi.sobel(g, axis=0, mode="nearest")
    assert np.allclose(np.abs(gx), np.abs(sx), atol=1e-3) and np.allclose(np.abs(gy), np.abs(sy), atol=1e-3)


def test_square_gives_one_closed_thin_contour():
    img = np.zeros((64, 64), np.uint8)
    img[16:48, 20:44] = 255
    e = app.canny(img, 100, 200)
    assert e.dtype == np.uint8 and set(np.unique(e)) <= {0, 255}
    on = e > 0
    assert not on[:12].any() and not on[52:].any() and not on[24:40, 26:38].any()          # flat regions: no edges
    rows, cols = np.nonzero(on)
    assert rows.min() in (15, 16) and rows.max() in (47, 48) and cols.min() in (19, 20) and cols.max() in (43, 44)
    perimeter = 2 * (32 + 24)
    assert 0.8 * perimeter <= on.sum() <= 2.2 * perimeter                                   # a thin (1-2 pixel) closed outline
    # every edge pixel has an edge neighbour: the contour is connected
    p = np.pad(on, 1)
    nb = sum(p[1 + dy:65 + dy, 1 + dx:65 + dx] for dy in (-1, 0, 1) for dx in (-1, 0, 1) if (dy, dx) != (0, 0))
    assert (nb[on] >= 1).all()


def test_hysteresis_keeps_weak_edges_only_when_connected_to_strong_ones():
    img = np.full((40, 80), 100, np.uint8)
    img[:, 20:] = 160                                  # a step of 60 grey levels: Sobel L1 magnitude 240 along x = 20
    img[10:30, 60:] = 180                              # a weaker, disconnected step of 20: magnitude 80
    strong_only = app.canny(img, 150, 200)
    assert strong_only[:, 18:22].any() and not strong_only[12:28, 57:63].any()
    both = app.canny(img, 50, 70)                      # low enough for the weak step to be strong on its own
    assert both[12:28, 57:63].any()
    linked = app.canny(img, 60, 200)                   # weak step above `low`, below `high`, NOT connected to a strong edge: dropped
    assert not linked[12:28, 57:63].any()


def test_raising_the_thresholds_never_adds_edges():
    rng = np.random.default_rng(1)
    img = (rng.random((48, 48)) * 255).astype(np.uint8)
    img[10:30, 10:30] = 255
    a, b, c = app.canny(img, 50, 100) > 0, app.canny(img, 50, 200) > 0, app.canny(img, 150, 200) > 0
    assert (b <= a).all() and (c <= b).all()


def test_colour_input_and_hwc3():
    img = np.zeros((32, 32, 3), np.uint8)
    img[8:24, 8:24] = (255, 0, 0)
    e = app.canny(img, 30, 60)
    assert e.shape == (32, 32) and e.any()
    assert app.hwc3(e).shape == (32, 32, 3)
    rgba = np.dstack([img, np.full((32, 32), 128, np.uint8)])
    assert app.hwc3(rgba).shape == (32, 32, 3)
