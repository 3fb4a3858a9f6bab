"""Device-side image input transform (pgca_image_preprocess) against the oracle restatement of the
reference's host path (oracle/image_restatement.py, itself pinned bit-exactly against Pillow in tests/test_image_cpu.py):
the resized uint8 image and the normalised float32 tensor must be BIT-EXACT."""
import numpy as np
import pytest
import torch

from oracle import image_restatement as IR

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rnd(b, h, w, seed):
    return np.random.RandomState(seed).randint(0, 256, (b, h, w, 3), dtype=np.uint8)


def want(imgs, S, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    res = np.stack([IR.resize_bilinear_u8_fast(im, S, S) for im in imgs])
    out = torch.stack([IR.to_tensor_normalize(r, mean, std) for r in res])
    return res, out


@pytest.mark.parametrize("b,h,w,s", [(3, 37, 53, 224), (2, 224, 224, 224), (2, 1, 1, 8), (1, 3, 500, 224),
                                     (2, 500, 375, 224), (1, 225, 223, 224), (1, 97, 1024, 224), (2, 768, 1024, 224),
                                     (5, 17, 19, 32), (1, 640, 480, 336), (3, 33, 21, 7), (2, 100, 60, 30),
                                     (1, 9, 4000, 224), (11, 5, 5, 224)])
def test_resize_normalize_bit_exact(b, h, w, s):
    from pgca_amd.input import GpuImageProcessor
    imgs = rnd(b, h, w, h * 131 + w)
    proc = GpuImageProcessor(s, device=DEV)
    out, res = proc.process_batch(torch.from_numpy(imgs), return_resized=True)
    wres, wout = want(imgs, s)
    assert res.dtype == torch.uint8 and np.array_equal(res.cpu().numpy(), wres)
    assert out.dtype == torch.float32 and out.shape == (b, 3, s, s)
    assert torch.equal(out.cpu(), wout)                      # bitwise: same IEEE divisions as torch on the host


def test_against_pillow_directly():
    PIL_Image = pytest.importorskip("PIL.Image")
    from pgca_amd.input import GpuImageProcessor
    img = rnd(1, 333, 517, 9)[0]
    pil = np.asarray(PIL_Image.fromarray(img, "RGB").resize((224, 224), PIL_Image.BILINEAR))
    _, res = GpuImageProcessor(224, device=DEV).process_batch(torch.from_numpy(img[None]), return_resized=True)
    assert np.array_equal(res[0].cpu().numpy(), pil)


def test_extremes_custom_statistics_and_unaligned_views():
    from pgca_amd.input import GpuImageProcessor
    proc = GpuImageProcessor(64, mean=(0.5, 0.25, 0.125), std=(0.5, 0.3, 0.7), device=DEV)
    for fill in (0, 255):
        imgs = np.full((2, 41, 29, 3), fill, np.uint8)
        out = proc.process_batch(torch.from_numpy(imgs))
        _, wout = want(imgs, 64, (0.5, 0.25, 0.125), (0.5, 0.3, 0.7))
        assert torch.equal(out.cpu(), wout)
    # a contiguous view whose first byte is not 16-byte aligned (41*29*3 = 3567 bytes per image)
    big = torch.from_numpy(rnd(4, 41, 29, 3)).to(DEV)
    view = big[1:]
    assert view.data_ptr() % 16 != 0
    out = proc.process_batch(view)
    _, wout = want(big[1:].cpu().numpy(), 64, (0.5, 0.25, 0.125), (0.5, 0.3, 0.7))
    assert torch.equal(out.cpu(), wout)


def test_mixed_size_list_and_errors():
    from pgca_amd.input import GpuImageProcessor
    proc = GpuImageProcessor(32, device=DEV)
    rs = np.random.RandomState(4)
    shapes = [(20, 30), (64, 48), (20, 30), (7, 90), (64, 48)]
    imgs = [rs.randint(0, 256, (h, w, 3), dtype=np.uint8) for h, w in shapes]
    out = proc([torch.from_numpy(i) for i in imgs])
    for i, im in enumerate(imgs):
        assert torch.equal(out[i].cpu(), IR.process_image(im, 32))
    with pytest.raises(ValueError, match="uint8 images"):
        proc.process_batch(torch.zeros(1, 8, 8, 3))
    with pytest.raises(ValueError, match="uint8 images"):
        proc.process_batch(torch.zeros(1, 3, 8, 8, dtype=torch.uint8))


# ------------------------------------------------------------------------------------------------ training transform
def _params(rs, H, W):
    h, w = int(rs.randint(H // 2, H + 1)), int(rs.randint(W // 2, W + 1))
    f = lambda lo, hi: float(np.float32(rs.uniform(lo, hi)))
    return dict(box=(int(rs.randint(0, H - h + 1)), int(rs.randint(0, W - w + 1)), h, w), flip=bool(rs.randint(0, 2)),
                order=[int(v) for v in rs.permutation(4)], brightness=f(0.8, 1.2), contrast=f(0.8, 1.2),
                saturation=f(0.8, 1.2), hue=f(-0.1, 0.1), angle=f(-5, 5))


@pytest.mark.parametrize("b,h,w,s", [(6, 256, 320, 224), (4, 97, 131, 224), (3, 500, 375, 224), (5, 64, 64, 32),
                                     (2, 224, 224, 224), (3, 300, 200, 230)])
def test_train_transform_bit_exact(b, h, w, s):
    """Reference train_transform (data/preprocessing.py:52-70) given the draws: the augmented uint8 image and the
    normalised tensor against the oracle (itself pinned to Pillow in tests/test_image_cpu.py)."""
    from pgca_amd.input import GpuImageProcessor
    rs = np.random.RandomState(h * 7 + w)
    imgs = rnd(b, h, w, h + 3 * w)
    params = [_params(rs, h, w) for _ in range(b)]
    params[0].update(angle=0.0, flip=True)                       # PIL's copy path of rotate
    params[-1].update(brightness=1.2, contrast=0.8, saturation=1.2, hue=-0.1, angle=5.0)
    proc = GpuImageProcessor(s, device=DEV)
    out, aug = proc.process_train_batch(torch.from_numpy(imgs), params, return_augmented=True)
    for i in range(b):
        want_u8 = IR.train_augment_u8(imgs[i], params[i], s)
        assert np.array_equal(aug[i].cpu().numpy(), want_u8), (i, params[i])
        assert torch.equal(out[i].cpu(), IR.to_tensor_normalize(want_u8))


def test_train_transform_against_pillow_directly():
    pytest.importorskip("PIL.Image")
    from pgca_amd.input import GpuImageProcessor
    rs = np.random.RandomState(12)
    imgs = rnd(4, 240, 300, 5)
    params = [_params(rs, 240, 300) for _ in range(4)]
    _, aug = GpuImageProcessor(224, device=DEV).process_train_batch(torch.from_numpy(imgs), params,
                                                                    return_augmented=True)
    for i in range(4):
        assert np.array_equal(aug[i].cpu().numpy(), IR.train_augment_u8_pil(imgs[i], params[i], 224))


def test_hue_turn_on_every_colour_and_blend_extremes():
    """All 2^24 colours through the device's HSV round trip (335 identity-cropped 224 x 224 images), and the three blends
    at their extreme factors on the same pixels."""
    from pgca_amd.input import GpuImageProcessor
    S, n = 224, 1 << 24
    B = (n + S * S - 1) // (S * S)
    v = np.arange(B * S * S, dtype=np.uint32) % n
    imgs = np.stack([(v >> 16) & 255, (v >> 8) & 255, v & 255], -1).astype(np.uint8).reshape(B, S, S, 3)
    proc = GpuImageProcessor(S, device=DEV)
    base = dict(box=(0, 0, S, S), flip=False, order=[0, 1, 2, 3], brightness=1.0, contrast=1.0, saturation=1.0, hue=0.0,
                angle=0.0)
    x = torch.from_numpy(imgs)
    for hue in (0.1, -0.1, 0.037):
        _, aug = proc.process_train_batch(x, [dict(base, hue=hue)] * B, return_augmented=True)
        assert np.array_equal(aug.cpu().numpy(), IR.adjust_hue(imgs, hue)), hue
    _, aug = proc.process_train_batch(x[:8], [dict(base, brightness=1.2, saturation=0.8)] * 8, return_augmented=True)
    # (the hue step runs the HSV round trip even with a zero turn, as torchvision's adjust_hue does)
    want = np.stack([IR.adjust_hue(IR.adjust_saturation(IR.adjust_brightness(im, 1.2), 0.8), 0.0) for im in imgs[:8]])
    assert np.array_equal(aug.cpu().numpy(), want)
    _, aug = proc.process_train_batch(x[:8], [dict(base, order=[1, 0, 2, 3], contrast=1.2, brightness=0.8)] * 8,
                                      return_augmented=True)
    want = np.stack([IR.adjust_hue(IR.adjust_brightness(IR.adjust_contrast(im, 1.2), 0.8), 0.0) for im in imgs[:8]])
    assert np.array_equal(aug.cpu().numpy(), want)


def test_train_transform_draws_and_errors():
    from pgca_amd.input import GpuImageProcessor, draw_train_params
    proc = GpuImageProcessor(64, device=DEV)
    imgs = torch.from_numpy(rnd(3, 90, 120, 2))
    a = proc.process_train_batch(imgs, generator=torch.Generator().manual_seed(7))
    b = proc.process_train_batch(imgs, generator=torch.Generator().manual_seed(7))
    assert torch.equal(a, b) and a.shape == (3, 3, 64, 64) and bool(torch.isfinite(a).all())
    g = torch.Generator().manual_seed(7)
    params = [draw_train_params(90, 120, g) for _ in range(3)]
    assert torch.equal(a, proc.process_train_batch(imgs, params))
    for i in range(3):
        assert torch.equal(a[i].cpu(), IR.process_image_train(imgs[i].numpy(), params[i], 64))
    with pytest.raises(ValueError, match="crop box"):
        proc.process_train_batch(imgs, [dict(params[0], box=(0, 0, 91, 10))] * 3)
    with pytest.raises(ValueError, match="permutation"):
        proc.process_train_batch(imgs, [dict(params[0], order=[0, 0, 1, 2])] * 3)
    with pytest.raises(ValueError, match="parameter sets"):
        proc.process_train_batch(imgs, params[:2])
    with pytest.raises(RuntimeError, match="LDS"):
        GpuImageProcessor(336, device=DEV).process_train_batch(imgs, params)


def test_prepare_takes_decoded_images():
    """``ContrastiveStep.prepare`` with a loader that hands over decoded uint8 images: the device transform (training
    with the given draws, validation otherwise) instead of a host-side float tensor."""
    from pgca_amd.steps import ContrastiveStep
    rs = np.random.RandomState(21)
    imgs = rnd(3, 80, 100, 8)
    params = [_params(rs, 80, 100) for _ in range(3)]
    ids = torch.randint(0, 100, (3, 16))
    mask = torch.ones(3, 16, dtype=torch.int64)
    batch = {"image": torch.from_numpy(imgs), "caption_ids": ids, "caption_mask": mask, "image_size": 32}
    val = ContrastiveStep.prepare(dict(batch), torch.device(DEV))["image"]
    tr = ContrastiveStep.prepare(dict(batch, augment=True, augment_params=params), torch.device(DEV))["image"]
    for i in range(3):
        assert torch.equal(val[i].cpu(), IR.process_image(imgs[i], 32))
        assert torch.equal(tr[i].cpu(), IR.process_image_train(imgs[i], params[i], 32))
    torch.manual_seed(3)
    a = ContrastiveStep.prepare(dict(batch, augment=True), torch.device(DEV))["image"]
    torch.manual_seed(3)
    b = ContrastiveStep.prepare(dict(batch, augment=True), torch.device(DEV))["image"]
    assert torch.equal(a, b) and not torch.equal(a, val)


def test_feeds_the_vision_tower():
    """The processed batch is what ``VisionEncoder.forward`` takes (reference model.py:210-218: 4-D, 3 channels)."""
    from pgca_amd.arch import tiny_arch
    from pgca_amd.input import GpuImageProcessor
    from pgca_amd.model import PreferenceGuidedCaptioningModel
    arch = tiny_arch()
    model = PreferenceGuidedCaptioningModel(freeze_vision_backbone=True, arch=arch, seed=2, device=DEV)
    imgs = rnd(2, 50, 70, 11)
    px = GpuImageProcessor(arch.vit.image, device=DEV).process_batch(torch.from_numpy(imgs))
    got = model.vision_encoder(px)["embeddings"]
    ref = model.vision_encoder(torch.stack([IR.process_image(i, arch.vit.image) for i in imgs]))["embeddings"]
    assert torch.equal(got, ref)
