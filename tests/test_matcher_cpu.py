"""Byte-exact image loading of the evaluation wrappers (reference: eval_tool/immatch/utils/data_io.py:48-62 =
cv2.imread(gray) -> cv2.resize on the UINT8 image -> to_tensor).  OpenCV is absent from the build container, so the
restatement of its published 8-bit paths is pinned by vectors derived BY HAND from the algorithm (the arithmetic of every
expected value is written out below), not by OpenCV's own output."""
import numpy as np

from geoformer_amd import matcher as MT


def test_gray_conversion_fixed_point():
    # (R*4899 + G*9617 + B*1868 + 8192) >> 14
    rgb = np.array([[[255, 255, 255], [0, 0, 0], [255, 0, 0], [0, 255, 0], [0, 0, 255], [10, 20, 30], [200, 100, 50]]], dtype=np.uint8)
    # white: 255*16384 + 8192 >> 14 = 255 ; red: (1249245 + 8192) >> 14 = 76 ; green: (2452335 + 8192) >> 14 = 150 ;
    # blue: (476340 + 8192) >> 14 = 29 ; (10,20,30): (48990 + 192340 + 56040 + 8192) >> 14 = 305562 >> 14 = 18 ;
    # (200,100,50): (979800 + 961700 + 93400 + 8192) >> 14 = 2043092 >> 14 = 124
    np.testing.assert_array_equal(MT.cv2_gray_u8(rgb), [[255, 0, 76, 150, 29, 18, 124]])


def test_resize_linear_upscale_row():
    # one row [10, 20] -> width 4: scale 0.5; fx = (dx + .5) * .5 - .5 = -.25, .25, .75, 1.25
    #   dx 0: sx = -1 -> clamped (sx 0, fx 0): 10 * 2048                          = 20480
    #   dx 1: sx 0, fx .25: 10 * 1536 + 20 * 512                                  = 25600
    #   dx 2: sx 0, fx .75: 10 * 512 + 20 * 1536                                  = 35840
    #   dx 3: sx 1 = w - 1 -> fx 0: 20 * 2048                                     = 40960
    # vertical (1 row -> 1 row): b = (2048, 0): ((2048 * (D >> 4)) >> 16) + 2 >> 2 = (40 + 2) >> 2 = 10 ; (50 + 2) >> 2 = 13 ;
    #   (70 + 2) >> 2 = 18 ; (80 + 2) >> 2 = 20     (12.5 -> 13 and 17.5 -> 18: round half up)
    np.testing.assert_array_equal(MT.cv2_resize_linear_u8(np.array([[10, 20]], dtype=np.uint8), 4, 1), [[10, 13, 18, 20]])


def test_resize_linear_two_dimensional():
    # [[0, 100], [200, 60]] -> 3 x 3: scale 2/3: f = -1/6, .5, 7/6 -> (s 0, f 0), (s 0, f .5), (s 1 clamped, f 0)
    #   horizontal rows (x 2048): row0 = [0, 0*1024 + 100*1024 = 102400, 204800] ; row1 = [409600, 200*1024 + 60*1024 = 266240, 122880]
    #   vertical: dy 0: fy = -1/6 -> sy -1, fy 5/6: rows clip(-1) = 0 and clip(0) = 0, b = (round(2048/6) = 341, round(2048*5/6) = 1707)
    #       x 1: r >> 4 = 6400: (341 * 6400 >> 16) + (1707 * 6400 >> 16) + 2 >> 2 = (33 + 166 + 2) >> 2 = 50
    #       x 2: r >> 4 = 12800: (66 + 333 + 2) >> 2 = 100 ; x 0: 0
    #   dy 1: fy .5: rows 0 and 1, b = (1024, 1024):
    #       x 0: (0 + (1024 * 25600 >> 16) + 2) >> 2 = (400 + 2) >> 2 = 100
    #       x 1: (1024 * 6400 >> 16) + (1024 * 16640 >> 16) + 2 >> 2 = (100 + 260 + 2) >> 2 = 90
    #       x 2: (1024 * 12800 >> 16) + (1024 * 7680 >> 16) + 2 >> 2 = (200 + 120 + 2) >> 2 = 80
    #   dy 2: fy = 7/6 -> sy 1, fy 1/6: rows clip(1) = 1 and clip(2) = 1, b = (1707, 341):
    #       x 0: r >> 4 = 25600: (1707 * 25600 >> 16) + (341 * 25600 >> 16) + 2 >> 2 = (666 + 133 + 2) >> 2 = 200
    #       x 1: r >> 4 = 16640: (433 + 86 + 2) >> 2 = 130 ; x 2: r >> 4 = 7680: (200 + 39 + 2) >> 2 = 60
    got = MT.cv2_resize_linear_u8(np.array([[0, 100], [200, 60]], dtype=np.uint8), 3, 3)
    np.testing.assert_array_equal(got, [[0, 50, 100], [100, 90, 80], [200, 130, 60]])


def test_resize_downscale_and_special_cases():
    src = (np.arange(8 * 12).reshape(8, 12) * 2 % 251).astype(np.uint8)
    np.testing.assert_array_equal(MT.cv2_resize_linear_u8(src, 12, 8), src)                       # same size: copy
    # exact 2x decimation: INTER_AREA, (a + b + c + d + 2) >> 2
    t = src.astype(int)
    np.testing.assert_array_equal(MT.cv2_resize_linear_u8(src, 6, 4), (t[0::2, 0::2] + t[0::2, 1::2] + t[1::2, 0::2] + t[1::2, 1::2] + 2) >> 2)
    # 4 -> 3 columns of a constant row stay constant (the two weights sum to 2048 for every fx: 2048 * v >> ... = v)
    c = np.full((5, 4), 137, dtype=np.uint8)
    np.testing.assert_array_equal(MT.cv2_resize_linear_u8(c, 3, 2), np.full((2, 3), 137))
    # a general downscale agrees with float bilinear interpolation (half-pixel centres) to 1 grey level
    rng = np.random.default_rng(5)
    im = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    got = MT.cv2_resize_linear_u8(im, 40, 24).astype(float)
    import torch
    ref = torch.nn.functional.interpolate(torch.from_numpy(im)[None, None].float(), size=(24, 40), mode='bilinear', align_corners=False)[0, 0].numpy()
    assert np.abs(got - ref).max() <= 1.0 and np.abs(got - ref).mean() < 0.3


def test_load_gray_scale_tensor_is_the_uint8_pipeline(tmp_path):
    from PIL import Image
    rng = np.random.default_rng(1)
    rgb = rng.integers(0, 256, (168, 200, 3), dtype=np.uint8)
    p = str(tmp_path / 'a.ppm')
    Image.fromarray(rgb).save(p)
    t, scale = MT.load_gray_scale_tensor(p, 'cpu', imsize=160, dfactor=8, value_to_scale=min)
    assert tuple(t.shape) == (1, 1, 160, 184) and scale == (200 / 184, 168 / 160)                # data_io.py:16-26
    want = MT.cv2_resize_linear_u8(MT.cv2_gray_u8(rgb), 184, 160)
    np.testing.assert_array_equal((t[0, 0].numpy() * 255).round().astype(np.uint8), want)          # k / 255 for integer k: to_tensor
    assert float((t * 255 - (t * 255).round()).abs().max()) < 1e-4


def test_jpeg_grayscale_is_the_decoders_luma_plane(tmp_path):
    """cv2.imread(path, IMREAD_GRAYSCALE) on a JPEG lets libjpeg emit the Y plane (JCS_GRAYSCALE); it is NOT the
    BGR2GRAY formula on the decoded RGB.  load_gray_image takes the same libjpeg path: its result equals channel 0 of a
    YCbCr decode of the file bit for bit, and differs from the fixed-point formula on the RGB decode."""
    from PIL import Image
    from geoformer_amd.matcher import cv2_gray_u8, load_gray_image
    rng = np.random.default_rng(7)
    base = rng.integers(0, 256, (12, 16, 3)).astype(np.uint8)
    rgb = np.kron(base, np.ones((8, 8, 1), np.uint8))                       # blocky colour image, 96 x 128
    rgb = np.clip(rgb.astype(np.int32) + rng.integers(-6, 7, rgb.shape), 0, 255).astype(np.uint8)
    path = str(tmp_path / 'pair.jpg')
    Image.fromarray(rgb).save(path, quality=90, subsampling=2)
    got = load_gray_image(path)
    ycc = Image.open(path)
    ycc.draft('YCbCr', ycc.size)
    luma = np.array(ycc)[..., 0]
    assert got.shape == (96, 128) and got.dtype == np.uint8
    np.testing.assert_array_equal(got, luma)
    formula = cv2_gray_u8(np.array(Image.open(path).convert('RGB'), dtype=np.uint8))
    assert int(np.abs(formula.astype(int) - got.astype(int)).max()) >= 1    # the two really are different functions
    assert float(np.abs(formula.astype(int) - got.astype(int)).mean()) < 1.0  # ... of nearly the same thing (RGB clipping: up to ~10 at saturated pixels)
    # a PNG of the same image still goes through the fixed-point formula
    png = str(tmp_path / 'pair.png')
    Image.fromarray(rgb).save(png)
    np.testing.assert_array_equal(load_gray_image(png), cv2_gray_u8(rgb))
