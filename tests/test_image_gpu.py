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
