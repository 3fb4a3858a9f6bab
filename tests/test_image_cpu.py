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
