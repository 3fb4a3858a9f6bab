"""The oracle of the image input transform (oracle/image_restatement.py) pinned against Pillow itself: the restated
two-pass fixed-point bilinear resample must reproduce ``PIL.Image.resize`` BIT-EXACTLY (reference data/preprocessing.py:44-48
reaches it through torchvision's ``Resize``), and the product's coefficient tables must equal the oracle's."""
import numpy as np
import pytest
import torch

from oracle import image_restatement as IR

PIL_Image = pytest.importorskip("PIL.Image")

CASES = [(37, 53, 224), (224, 224, 224), (1, 1, 8), (3, 500, 224), (500, 375, 224), (640, 480, 224), (225, 223, 224),
         (97, 1024, 224), (768, 1024, 224), (17, 19, 32), (2, 2, 7)]


def rnd(h, w, seed):
    return np.random.RandomState(seed).randint(0, 256, (h, w, 3), dtype=np.uint8)


@pytest.mark.parametrize("h,w,s", CASES)
def test_resample_matches_pillow_bit_exactly(h, w, s):
    img = rnd(h, w, h * 1000 + w)
    want = np.asarray(PIL_Image.fromarray(img, "RGB").resize((s, s), PIL_Image.BILINEAR))
    got = IR.resize_bilinear_u8_fast(img, s, s)
    assert got.dtype == np.uint8 and np.array_equal(got, want)
    if h * w <= 64 * 64:    # the plain-loop statement of the C code, small cases only
        assert np.array_equal(IR.resize_bilinear_u8(img, s, s), want)


def test_extreme_pixels_and_non_square_targets():
    for fill in (0, 255):
        img = np.full((50, 70, 3), fill, np.uint8)
        assert np.array_equal(IR.resize_bilinear_u8_fast(img, 224, 224), np.full((224, 224, 3), fill, np.uint8))
    img = rnd(60, 90, 5)
    want = np.asarray(PIL_Image.fromarray(img, "RGB").resize((48, 32), PIL_Image.BILINEAR))   # PIL takes (width, height)
    assert np.array_equal(IR.resize_bilinear_u8_fast(img, 32, 48), want)


def test_to_tensor_normalize_definition():
    img = rnd(8, 8, 1)
    t = IR.to_tensor_normalize(img)
    assert t.shape == (3, 8, 8) and t.dtype == torch.float32
    c, y, x = 1, 3, 5
    want = (np.float32(img[y, x, c]) / np.float32(255)) - np.float32(0.456)
    want = np.float32(want) / np.float32(0.224)
    assert float(t[c, y, x]) == float(want)


@pytest.mark.parametrize("n_in,n_out", [(53, 224), (224, 224), (1024, 224), (375, 224), (1, 8), (500, 7)])
def test_product_coefficient_tables_equal_the_oracle(n_in, n_out):
    from pgca_amd.input import resample_tables
    b, k, ks = IR.precompute_coeffs(n_in, n_out)
    pb, pk = resample_tables(n_in, n_out)
    assert pk.shape == (n_out, ks) and pk.dtype == np.int32 and pb.dtype == np.int32
    assert np.array_equal(pb, b) and np.array_equal(pk, k)


# ------------------------------------------------------------------------------------------------ training transform
# (reference data/preprocessing.py:52-70).  Every restated pixel operation against Pillow ITSELF, exhaustively where the
# domain is small enough (all 2^24 colours for the colour-space conversions, all 2^16 value pairs for the blend).
def _all_colours(lo, hi):
    v = np.arange(lo, hi, dtype=np.uint32)
    return np.stack([(v >> 16) & 255, (v >> 8) & 255, v & 255], -1).astype(np.uint8)


def test_rgb_hsv_round_trip_matches_pillow_on_every_colour():
    step = 1 << 20
    for lo in range(0, 1 << 24, step):
        px = _all_colours(lo, lo + step).reshape(1024, 1024, 3)
        assert np.array_equal(IR.rgb2hsv(px), np.asarray(PIL_Image.fromarray(px, "RGB").convert("HSV")))
        assert np.array_equal(IR.hsv2rgb(px), np.asarray(PIL_Image.fromarray(px, "HSV").convert("RGB")))
        assert np.array_equal(IR.rgb2l(px), np.asarray(PIL_Image.fromarray(px, "RGB").convert("L")))


def test_blend_matches_pillow_on_every_value_pair():
    a = np.arange(256, dtype=np.uint8)
    in1, in2 = np.repeat(a[:, None], 256, 1), np.repeat(a[None, :], 256, 0)
    rs = np.random.RandomState(0)
    alphas = [0.0, 1.0, 0.5, 0.8, 1.2] + [float(np.float32(x)) for x in rs.uniform(0.7, 1.3, 100)]
    for al in alphas:
        want = np.asarray(PIL_Image.blend(PIL_Image.fromarray(in1, "L"), PIL_Image.fromarray(in2, "L"), al))
        assert np.array_equal(IR.blend_u8(in1, in2, al), want), al


def test_enhancers_hue_and_rotation_match_pillow():
    from PIL import ImageEnhance
    rs = np.random.RandomState(1)
    for h, w in [(64, 48), (33, 97), (224, 224)]:
        img = rnd(h, w, h + w)
        pil = PIL_Image.fromarray(img, "RGB")
        for f in [0.8, 1.2, 1.0] + [float(np.float32(x)) for x in rs.uniform(0.8, 1.2, 6)]:
            assert np.array_equal(IR.adjust_brightness(img, f), np.asarray(ImageEnhance.Brightness(pil).enhance(f)))
            assert np.array_equal(IR.adjust_contrast(img, f), np.asarray(ImageEnhance.Contrast(pil).enhance(f)))
            assert np.array_equal(IR.adjust_saturation(img, f), np.asarray(ImageEnhance.Color(pil).enhance(f)))
        for ang in [0.0, 5.0, -5.0, 0.001, -0.001, 360.0] + [float(np.float32(a)) for a in rs.uniform(-5, 5, 20)]:
            want = np.asarray(pil.rotate(ang, PIL_Image.NEAREST, False, None, fillcolor=0))
            assert np.array_equal(IR.rotate_nearest(img, ang), want), ang


def _params(rs, H, W):
    h, w = int(rs.randint(H // 2, H + 1)), int(rs.randint(W // 2, W + 1))
    f = lambda lo, hi: float(np.float32(rs.uniform(lo, hi)))
    return dict(box=(int(rs.randint(0, H - h + 1)), int(rs.randint(0, W - w + 1)), h, w), flip=bool(rs.randint(0, 2)),
                order=[int(v) for v in rs.permutation(4)], brightness=f(0.8, 1.2), contrast=f(0.8, 1.2),
                saturation=f(0.8, 1.2), hue=f(-0.1, 0.1), angle=f(-5, 5))


def test_whole_training_chain_matches_pillow():
    rs = np.random.RandomState(2)
    for k in range(12):
        H, W = int(rs.randint(60, 300)), int(rs.randint(60, 300))
        img, p = rnd(H, W, k), _params(rs, H, W)
        assert np.array_equal(IR.train_augment_u8(img, p, 64), IR.train_augment_u8_pil(img, p, 64)), p
    img, p = rnd(256, 320, 99), _params(rs, 256, 320)
    a = IR.train_augment_u8(img, p, 224)
    assert np.array_equal(a, IR.train_augment_u8_pil(img, p, 224))
    t = IR.process_image_train(img, p, 224)
    assert t.shape == (3, 224, 224) and torch.equal(t, IR.to_tensor_normalize(a))


def test_train_parameter_draws_and_rotation_coefficients():
    from pgca_amd.input import draw_train_params, rotate_fixed_coeffs
    g = torch.Generator().manual_seed(5)
    seen_flip = set()
    for _ in range(200):
        p = draw_train_params(256, 320, g)
        i, j, h, w = p["box"]
        assert 0 <= i and 0 <= j and 0 < h and 0 < w and i + h <= 256 and j + w <= 320
        assert 0.8 * 256 * 320 * 0.97 <= h * w <= 256 * 320             # scale (0.8, 1.0), up to integer rounding
        assert 0.74 <= w / h <= 1.35                                    # ratio (0.75, 1.33)
        assert sorted(p["order"]) == [0, 1, 2, 3]
        assert 0.8 <= p["brightness"] <= 1.2 and 0.8 <= p["contrast"] <= 1.2 and 0.8 <= p["saturation"] <= 1.2
        assert -0.1 <= p["hue"] <= 0.1 and -5.0 <= p["angle"] <= 5.0
        seen_flip.add(p["flip"])
    assert seen_flip == {True, False}
    # same generator state -> same draws; an extreme aspect ratio falls back to the central crop
    a = draw_train_params(100, 100, torch.Generator().manual_seed(3))
    assert a == draw_train_params(100, 100, torch.Generator().manual_seed(3))
    p = draw_train_params(10, 200, torch.Generator().manual_seed(0))
    assert p["box"] == (0, 93, 10, 13)                                  # in_ratio 20 > 1.33: h = 10, w = round(13.3)
    for ang in (0.0, 360.0, 3.3, -4.9):
        assert rotate_fixed_coeffs(ang, 224, 224) == IR.rotate_fixed_coeffs(ang, 224, 224)
    assert rotate_fixed_coeffs(0.0, 224, 224) is None
