"""Pins the oracle's colour-encoding pieces on the literal vectors of the reference's own unit tests (numbers copied as data):
  Application/Tests/test_pixels.cpp:629-795    vec_to_r3g3b2 / r3g3b2_to_vec / convert_to_r3g3b2 / convert_from_r3g3b2
  Application/Tests/test_pixels.cpp:915-1071   LineWithoutGridTest2, r3g3b2 and rgb8 rows (gray background 100 / rgb8 background 100)
  Application/Tests/test_pixels.cpp:1073-1166  BackgroundThresholding.RGB8AbsoluteDifferenceSimulatedBlob (rgb8 leg)
  Application/Tests/test_pixels.cpp:1289-1380  BlobThresholding.RGB8AbsoluteDifferenceMultiRow
  Application/Tests/test_pixels.cpp:1531-1577  ImageFromLines.RGB8BackgroundSubtractionUsesAllChannels
HorizontalLine literals there are (y, x0, x1); colour triples are in memory order."""
import numpy as np
from oracle import oracle
from test_oracle_golden import R, lines, ABS, SIGN, NONE

FULL = R((0, 0, 9), (1, 0, 9))


def test_r3g3b2_bit_layout():
    assert oracle.vec_to_r3g3b2(255, 128, 64) == 0b11100010
    assert oracle.r3g3b2_to_vec(0b11100010).tolist() == [192, 128, 64]
    assert oracle.vec_to_r3g3b2(255, 0, 0) == 0b11000000
    assert oracle.vec_to_r3g3b2(0, 255, 0) == 0b00111000
    assert oracle.vec_to_r3g3b2(0, 0, 255) == 0b00000111
    assert oracle.vec_to_r3g3b2(255, 255, 255) == 0xff and oracle.vec_to_r3g3b2(0, 0, 0) == 0
    assert oracle.r3g3b2_to_vec(0b11000000).tolist() == [192, 0, 0]
    assert oracle.r3g3b2_to_vec(0b00111000).tolist() == [0, 224, 0]
    assert oracle.r3g3b2_to_vec(0b00000111).tolist() == [0, 0, 224]
    assert oracle.r3g3b2_to_vec(0xff).tolist() == [192, 224, 224]
    img = np.zeros((2, 2, 3), np.uint8); img[...] = (255, 128, 64)
    assert np.all(oracle.convert_to_r3g3b2(img) == 0b11100010)
    img4 = np.zeros((2, 2, 4), np.uint8); img4[...] = (255, 128, 64, 255)
    assert np.all(oracle.convert_to_r3g3b2(img4) == 0b11100010)
    row = np.array([[(255, 0, 0), (0, 255, 0), (0, 0, 255)]], np.uint8)
    assert oracle.convert_to_r3g3b2(row).tolist() == [[0b11000000, 0b00111000, 0b00000111]]


def test_line_without_grid2_r3g3b2_rows():
    bg = np.full((10, 10), 100, np.uint8)
    px = np.array([oracle.vec_to_r3g3b2(i * 10, i * 10, i * 10) for i in range(20)], np.uint8)
    r, p = oracle.line_without_grid_enc(FULL, px, oracle.ENC_R3G3B2, bg, oracle.ENC_GRAY, ABS, 50)
    assert lines(r) == [(0, 0, 6), (1, 6, 9)]
    assert p.tolist() == [0, 0, 0, 0, 9, 9, 9, 173, 173, 173, 173]
    r, p = oracle.line_without_grid_enc(FULL, px, oracle.ENC_R3G3B2, bg, oracle.ENC_GRAY, SIGN, 50)
    assert lines(r) == [(0, 0, 6)] and p.tolist() == [0, 0, 0, 0, 9, 9, 9]
    r, p = oracle.line_without_grid_enc(FULL, px, oracle.ENC_R3G3B2, bg, oracle.ENC_GRAY, NONE, 50)
    assert lines(r) == [(0, 7, 9), (1, 0, 9)]
    assert p.tolist() == [82, 82, 82, 91, 91, 91, 164, 164, 164, 173, 173, 173, 173]


def test_line_without_grid2_rgb8_rows():
    bg = np.full((10, 10, 3), 100, np.uint8)
    px = np.repeat((np.arange(20) * 10).astype(np.uint8), 3)
    r, p = oracle.line_without_grid_enc(FULL, px, oracle.ENC_RGB8, bg, oracle.ENC_RGB8, ABS, 50)
    assert lines(r) == [(0, 0, 5), (1, 5, 9)]
    assert p.tolist() == [v for v in [0, 10, 20, 30, 40, 50, 150, 160, 170, 180, 190] for _ in range(3)]
    r, p = oracle.line_without_grid_enc(FULL, px, oracle.ENC_RGB8, bg, oracle.ENC_RGB8, SIGN, 50)
    assert lines(r) == [(0, 0, 5)] and p.tolist() == [v for v in [0, 10, 20, 30, 40, 50] for _ in range(3)]
    r, p = oracle.line_without_grid_enc(FULL, px, oracle.ENC_RGB8, bg, oracle.ENC_RGB8, NONE, 50)
    assert lines(r) == [(0, 5, 9), (1, 0, 9)] and p.tolist() == [v for v in range(50, 200, 10) for _ in range(3)]


def _bg_4x2():
    bgv = np.array([[30, 50, 70, 90], [40, 60, 80, 100]], np.uint8)
    return np.repeat(bgv[:, :, None], 3, axis=2)


def test_rgb8_absolute_difference_simulated_blob():
    for last in ((100, 100, 100), (90, 90, 90)):             # :1073-1166 and :1289-1380 differ in the last pixel only
        blob = [(25, 25, 25), (110, 110, 110), (80, 80, 80), (10, 200, 10), (30, 30, 30), (95, 95, 95), (200, 200, 200), last]
        px = np.array(blob, np.uint8).reshape(-1)
        r, p = oracle.line_without_grid_enc(R((0, 0, 3), (1, 0, 3)), px, oracle.ENC_RGB8, _bg_4x2(), oracle.ENC_RGB8, ABS, 25)
        assert lines(r) == [(0, 1, 1), (0, 3, 3), (1, 1, 2)]
        assert p.tolist() == [110, 110, 110, 10, 200, 10, 95, 95, 95, 200, 200, 200]
        # the gray leg of the same tests (cv::cvtColor background, cmn::bgr2gray pixels) gives the same lines
        g = np.array([oracle.lib().oracle_bgr2gray(*v) for v in blob], np.uint8)
        r2, p2 = oracle.line_without_grid(R((0, 0, 3), (1, 0, 3)), g, _bg_4x2()[:, :, 0], ABS, 25)
        assert lines(r2) == lines(r)
        assert p2.tolist() == [oracle.lib().oracle_bgr2gray(*v) for v in [(110, 110, 110), (10, 200, 10), (95, 95, 95), (200, 200, 200)]]


def test_rgb8_background_subtraction_uses_all_channels():
    bg = np.full((1, 1, 3), 10, np.uint8)
    for blob in [(200, 10, 10), (10, 200, 10), (10, 10, 200), (200, 200, 200)]:
        r, p = oracle.line_without_grid_enc(R((0, 0, 0)), np.array(blob, np.uint8), oracle.ENC_RGB8, bg, oracle.ENC_RGB8, ABS, 20)
        assert lines(r) == [(0, 0, 0)] and p.tolist() == list(blob)


def test_rgb8_difference_image_reference_vector():
    # ImageFromLines.RGB8AbsoluteThresholdWithBackground (Tests/test_pixels.cpp:1381-1479): the `differences` image of an rgb8 blob
    # against an rgb8 background is the per-channel absolute difference of the pixels that pass the threshold
    blob = np.array([(25, 25, 25), (110, 110, 110), (80, 80, 80), (10, 200, 10), (30, 30, 30), (95, 95, 95), (200, 200, 200), (100, 100, 100)], np.uint8)
    frame = blob.reshape(2, 4, 3)
    bgc = _bg_4x2()
    r, _ = oracle.line_without_grid_enc(R((0, 0, 3), (1, 0, 3)), blob.reshape(-1), oracle.ENC_RGB8, bgc, oracle.ENC_RGB8, ABS, 25)
    kept = np.zeros(1, oracle.BLOB_DTYPE)[0]
    kept["x0"], kept["y0"], kept["x1"], kept["y1"], kept["run_begin"], kept["n_runs"] = 0, 0, 3, 1, 0, len(r)
    diff = np.stack([oracle.crop_none(frame[..., c], bgc[..., c], kept, r, out_w=4, out_h=2, difference=1) for c in range(3)], axis=-1)
    expected = np.array([[(0, 0, 0), (60, 60, 60), (0, 0, 0), (80, 110, 80)],
                         [(0, 0, 0), (35, 35, 35), (120, 120, 120), (0, 0, 0)]], np.uint8)
    assert np.array_equal(diff, expected)
    image = np.stack([oracle.crop_none(frame[..., c], bgc[..., c], kept, r, out_w=4, out_h=2) for c in range(3)], axis=-1)
    expected_image = np.array([[(0, 0, 0), (110, 110, 110), (0, 0, 0), (10, 200, 10)],
                               [(0, 0, 0), (95, 95, 95), (200, 200, 200), (0, 0, 0)]], np.uint8)
    assert np.array_equal(image, expected_image)
