"""Input path of the step loop: host batch -> pinned staging -> asynchronous H2D copy -> index
preparation ON THE DEVICE (``pgca_seq_batch_prepare``), ``depth`` batches ahead of the step that consumes them.

The reference feeds its loop from a ``DataLoader`` with ``pin_memory`` and moves tensors with Accelerate
(reference data/loader.py:533-572, training/trainer.py:464,575); the batch-dict contract (loader.py:252-258,487-497)
is kept - ``prepare`` is ``DPOStep.prepare`` / ``ContrastiveStep.prepare``.  At > 1000 pairs/s per GPU a synchronous
``.to(device)`` + host-side ``nonzero`` per micro-batch would sit on the critical path; here the copy engine and a side
HIP stream do that work under the previous step's kernels, and the only host wait (the number of scored rows, needed to
size the LM-head launch) happens in the feeder thread.
"""
from __future__ import annotations

import math
import queue
import threading
from typing import Callable, Iterable, Iterator

import torch

_STOP = object()

PRECISION_BITS = 32 - 8 - 2     # Pillow src/libImaging/Resample.c


def resample_tables(in_size: int, out_size: int):
    """Tap tables of Pillow's antialiased bilinear resample for ``in_size -> out_size`` over the whole image
    (Resample.c ``precompute_coeffs`` with the bilinear filter, support 1, then ``normalize_coeffs_8bpc``): returns
    ``(bounds int32 [out, 2] = (first tap, tap count), coef int32 [out, ksize])``, 22-bit fixed point.  Every float64
    operation is done in the C code's order (sequential tap sum, divide, scale, round half away from zero by truncation) so
    the integers are Pillow's own; ``tests/test_image_cpu.py`` holds them equal to the oracle's."""
    import numpy as np
    if in_size <= 0 or out_size <= 0:
        raise ValueError("sizes must be positive")
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    xx = np.arange(out_size, dtype=np.float64)
    center = 0.0 + (xx + 0.5) * scale
    ss = 1.0 / filterscale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)            # C (int) truncation; values are >= -0.5
    xmin = np.where(center - support + 0.5 < 0, 0, xmin)
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size) - xmin
    k = np.zeros((out_size, ksize), np.float64)
    ww = np.zeros(out_size, np.float64)
    for x in range(ksize):
        arg = (x + xmin - center + 0.5) * ss
        w = np.where(np.abs(arg) < 1.0, 1.0 - np.abs(arg), 0.0)
        w = np.where(x < xmax, w, 0.0)
        k[:, x] = w
        ww = ww + w                                                            # sequential, as the C loop
    nz = ww != 0.0
    k[nz] = k[nz] / ww[nz, None]
    fixed = np.where(k < 0, -0.5 + k * (1 << PRECISION_BITS), 0.5 + k * (1 << PRECISION_BITS)).astype(np.int64)
    bounds = np.stack([xmin, xmax], axis=1).astype(np.int32)
    return bounds, fixed.astype(np.int32)


class GpuImageProcessor:
    """The reference's ``ImageProcessor.val_transform`` (data/preprocessing.py:44-48,78; also its training transform with
    ``augment=False``, :70) on the device: decoded RGB uint8 ``[B, H, W, 3]`` (or a list of ``[H, W, 3]`` images of any
    sizes) -> ``Resize((S, S))`` -> ``ToTensor`` -> ``Normalize(mean, std)`` -> f32 ``[B, 3, S, S]``, bit-exact with the host
    path (``pgca_image_preprocess``).  ``process_train_batch`` is the training transform WITH the random augmentations
    (preprocessing.py:53-68): the draws are made on the host, the pixels are worked on the device."""

    def __init__(self, image_size: int = 224, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225), device=None):
        import numpy as np
        self.image_size = int(image_size)
        # float32 statistics, as torchvision's Normalize builds them (torch.as_tensor(mean, dtype=float32))
        self.mean = tuple(float(np.float32(m)) for m in mean)
        self.std = tuple(float(np.float32(s)) for s in std)
        if any(s == 0 for s in self.std):
            raise ValueError("std evaluated to zero after conversion to float32, leading to division by zero.")
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self._tables = {}
        self._host_tables = {}

    def _table(self, n_in: int):
        t = self._tables.get(n_in)
        if t is None:
            b, k = resample_tables(n_in, self.image_size)
            t = (torch.from_numpy(b).to(self.device), torch.from_numpy(k).to(self.device))
            self._tables[n_in] = t
        return t

    def process_batch(self, images: torch.Tensor, out: torch.Tensor = None, return_resized: bool = False):
        """images uint8 [B, H, W, 3] (host or device) -> f32 [B, 3, S, S] on the device."""
        from . import hip
        if images.dtype != torch.uint8 or images.dim() != 4 or images.shape[-1] != 3:
            raise ValueError(f"Expected uint8 images of shape (B, H, W, 3), got {images.dtype} {tuple(images.shape)}")
        x = images.to(self.device, non_blocking=True).contiguous()
        B, H, W, _ = x.shape
        S = self.image_size
        xb, xk = self._table(W)
        yb, yk = self._table(H)
        tmp = torch.empty((B, H, S, 3), dtype=torch.uint8, device=self.device)
        if out is None:
            out = torch.empty((B, 3, S, S), dtype=torch.float32, device=self.device)
        res = torch.empty((B, S, S, 3), dtype=torch.uint8, device=self.device) if return_resized else None
        hip.image_preprocess(x, B, H, W, S, xb, xk, yb, yk, self.mean, self.std, tmp, out, resized_u8=res)
        return (out, res) if return_resized else out

    def _batched_tables(self, sizes):
        """Tap tables of ``size_b -> S`` for every image of the batch, rows zero-padded to the widest one."""
        import numpy as np
        S = self.image_size
        tabs = []
        for n in sizes:
            t = self._host_tables.get(n)
            if t is None:
                t = self._host_tables[n] = resample_tables(n, S)
            tabs.append(t)
        kmax = max(t[1].shape[1] for t in tabs)
        bounds = np.stack([t[0] for t in tabs])
        coef = np.zeros((len(tabs), S, kmax), np.int32)
        for b, t in enumerate(tabs):
            coef[b, :, :t[1].shape[1]] = t[1]
        return torch.from_numpy(bounds).to(self.device), torch.from_numpy(coef).to(self.device)

    def process_train_batch(self, images: torch.Tensor, params=None, generator: torch.Generator = None,
                            out: torch.Tensor = None, return_augmented: bool = False):
        """The reference's ``train_transform`` with ``augment=True`` (data/preprocessing.py:52-70) on the device:
        RandomResizedCrop -> RandomHorizontalFlip -> ColorJitter -> RandomRotation(5) -> ToTensor -> Normalize, bit-exact
        with torchvision's PIL backend given the draws.  ``images`` uint8 [B, H, W, 3]; ``params``: one dict per image
        as ``draw_train_params`` returns them (drawn here from ``generator`` when omitted).  Returns f32 [B, 3, S, S]
        (and the augmented uint8 images [B, S, S, 3] with ``return_augmented``)."""
        import numpy as np
        from . import hip
        if images.dtype != torch.uint8 or images.dim() != 4 or images.shape[-1] != 3:
            raise ValueError(f"Expected uint8 images of shape (B, H, W, 3), got {images.dtype} {tuple(images.shape)}")
        B, H, W, _ = images.shape
        S = self.image_size
        if params is None:
            params = [draw_train_params(H, W, generator) for _ in range(B)]
        if len(params) != B:
            raise ValueError(f"{len(params)} parameter sets for {B} images")
        ints = np.zeros((B, TRAIN_PARAM_INTS), np.int32)
        fac = np.zeros((B, 3), np.float32)
        for b, p in enumerate(params):
            i, j, h, w = (int(v) for v in p["box"])
            if not (0 <= i and 0 <= j and 0 < h and 0 < w and i + h <= H and j + w <= W):
                raise ValueError(f"crop box {p['box']} outside the {H} x {W} image")
            order = [int(v) for v in p["order"]]
            if sorted(order) != [0, 1, 2, 3]:
                raise ValueError(f"order must be a permutation of 0..3, got {p['order']}")
            rot = rotate_fixed_coeffs(float(p["angle"]), S, S)
            ints[b, :4] = (i, j, h, w)
            ints[b, 4] = 1 if p["flip"] else 0
            ints[b, 5:9] = order
            ints[b, 9] = int(float(p["hue"]) * 255) % 256     # torchvision: np.array(hue * 255).astype(np.uint8)
            if rot is not None:
                ints[b, 10] = 1
                ints[b, 11:17] = rot
            fac[b] = (p["brightness"], p["contrast"], p["saturation"])
        x = images.to(self.device, non_blocking=True).contiguous()
        xb, xk = self._batched_tables([int(r[3]) for r in ints])
        yb, yk = self._batched_tables([int(r[2]) for r in ints])
        dev = self.device
        tmp = torch.empty((B, H, S, 3), dtype=torch.uint8, device=dev)
        res = torch.empty((B, S, S, 3), dtype=torch.uint8, device=dev)
        aug = torch.empty((B, S, S, 3), dtype=torch.uint8, device=dev) if return_augmented else None
        if out is None:
            out = torch.empty((B, 3, S, S), dtype=torch.float32, device=dev)
        hip.image_train_transform(x, B, H, W, S, torch.from_numpy(ints).to(dev), torch.from_numpy(fac).to(dev), xb, xk, yb,
                                  yk, self.mean, self.std, tmp, res, out, aug_u8=aug)
        return (out, aug) if return_augmented else out

    def __call__(self, images):
        """A uint8 batch tensor, or a list of uint8 ``[H, W, 3]`` images of mixed sizes (grouped by size, order kept)."""
        if isinstance(images, torch.Tensor):
            return self.process_batch(images if images.dim() == 4 else images[None])
        S = self.image_size
        out = torch.empty((len(images), 3, S, S), dtype=torch.float32, device=self.device)
        groups = {}
        for i, im in enumerate(images):
            groups.setdefault(tuple(im.shape), []).append(i)
        for idx in groups.values():
            batch = torch.stack([images[i] for i in idx])
            res = self.process_batch(batch)
            out[torch.as_tensor(idx, device=self.device)] = res
        return out


# ----------------------------------------------------------------------------------------------- training transform
TRAIN_PARAM_INTS = 20       # include/pgca_hip.h pgca_image_train_transform


def _fix16(v: float) -> int:
    v = v * 65536.0 + 0.5                      # Pillow Geometry.c FIX = FLOOR(v * 65536.0 + 0.5)
    return int(math.floor(v)) if v < 0 else int(v)


def rotate_fixed_coeffs(angle: float, w: int, h: int):
    """``PIL.Image.rotate(angle, NEAREST, expand=False, center=None)``: the output->input affine matrix as Pillow builds it
    (cos / sin rounded to 15 digits, centre (w/2, h/2)) turned into Geometry.c ``affine_fixed``'s six 16.16 integers
    (a0, a1, a2, a3, a4, a5); None when ``angle % 360 == 0`` (Pillow returns a copy)."""
    angle = angle % 360.0
    if angle == 0:
        return None
    cx, cy = w / 2.0, h / 2.0
    a = -math.radians(angle)
    m = [round(math.cos(a), 15), round(math.sin(a), 15), 0.0, round(-math.sin(a), 15), round(math.cos(a), 15), 0.0]
    m[2], m[5] = m[0] * -cx + m[1] * -cy + m[2], m[3] * -cx + m[4] * -cy + m[5]
    m[2] += cx
    m[5] += cy
    return (_fix16(m[0]), _fix16(m[1]), _fix16(m[2] + m[0] * 0.5 + m[1] * 0.5),
            _fix16(m[3]), _fix16(m[4]), _fix16(m[5] + m[3] * 0.5 + m[4] * 0.5))


def draw_train_params(height: int, width: int, generator: torch.Generator = None, scale=(0.8, 1.0), ratio=(0.75, 1.33),
                      flip_p: float = 0.5, brightness: float = 0.2, contrast: float = 0.2, saturation: float = 0.2,
                      hue: float = 0.1, degrees: float = 5.0) -> dict:
    """One image's random draws of the reference's training transform (data/preprocessing.py:53-68), made with the torch
    RNG calls torchvision makes and in its order: ``RandomResizedCrop.get_params`` (up to ten (area, log-ratio) tries, then
    the central-crop fallback), ``RandomHorizontalFlip`` (``torch.rand(1) < p``), ``ColorJitter.get_params``
    (``randperm(4)`` then the four uniform factors), ``RandomRotation.get_params``.  torchvision is not installed here:
    restated from its published source, so the DRAW ORDER is unpinned; the pixel work given these draws is pinned
    against Pillow (tests/test_image_cpu.py, tests/test_image_gpu.py)."""
    g = generator

    def uniform(lo, hi):
        return float(torch.empty(1).uniform_(float(lo), float(hi), generator=g))

    area = height * width
    log_ratio = torch.log(torch.tensor(ratio))
    box = None
    for _ in range(10):
        target_area = area * uniform(scale[0], scale[1])
        aspect = float(torch.exp(torch.empty(1).uniform_(float(log_ratio[0]), float(log_ratio[1]), generator=g)))
        w = int(round(math.sqrt(target_area * aspect)))
        h = int(round(math.sqrt(target_area / aspect)))
        if 0 < w <= width and 0 < h <= height:
            i = int(torch.randint(0, height - h + 1, size=(1,), generator=g))
            j = int(torch.randint(0, width - w + 1, size=(1,), generator=g))
            box = (i, j, h, w)
            break
    if box is None:                                     # fallback to the central crop
        in_ratio = float(width) / float(height)
        if in_ratio < min(ratio):
            w = width
            h = int(round(w / min(ratio)))
        elif in_ratio > max(ratio):
            h = height
            w = int(round(h * max(ratio)))
        else:
            w, h = width, height
        box = ((height - h) // 2, (width - w) // 2, h, w)
    flip = bool(torch.rand(1, generator=g) < flip_p)
    order = [int(v) for v in torch.randperm(4, generator=g)]
    b = uniform(max(0.0, 1.0 - brightness), 1.0 + brightness)
    c = uniform(max(0.0, 1.0 - contrast), 1.0 + contrast)
    sfac = uniform(max(0.0, 1.0 - saturation), 1.0 + saturation)
    hfac = uniform(-hue, hue)
    angle = uniform(-degrees, degrees)
    return {"box": box, "flip": flip, "order": order, "brightness": b, "contrast": c, "saturation": sfac, "hue": hfac,
            "angle": angle}


class TokenisedCaptionCache:
    """Tokenised-caption cache (reference ``TextProcessor.encode_caption``, data/preprocessing.py:206-238,
    called once per sample per epoch from ``ConceptualCaptionsDataset.__getitem__`` / ``UltraFeedbackDataset.__getitem__``,
    data/loader.py:239-258,470-497).  The reference re-runs the tokenizer for the same caption every epoch; here every distinct
    caption is encoded ONCE into a row of a pinned ``[capacity, max_length]`` int64 table (ids) and its int64 mask table, and
    batches are gathered by row index - a memcpy into staging that the H2D copy engine can take directly.

    ``tokenizer``: anything callable the way the reference calls HF's (``tokenizer(caption, max_length=, padding=,
    truncation=, add_special_tokens=, return_tensors="pt", return_attention_mask=)`` -> mapping with ``input_ids`` /
    ``attention_mask`` of shape [1, L]).  ``encode_caption`` returns exactly what the reference's method returns
    (``{"input_ids": [max_length], "attention_mask": [max_length]}`` for ``padding="max_length"``); rows shorter than
    ``max_length`` (a ``padding="longest"`` tokenizer) are right-padded with ``pad_token_id`` and mask 0, the layout the step
    kernels' packed rows assume.  ``save`` / ``load`` persist the table (captions hashed, not stored)."""

    def __init__(self, tokenizer, max_length: int = 128, padding: str = "max_length", truncation: bool = True,
                 capacity: int = 1024, pin: bool = True):
        self.tokenizer, self.max_length, self.padding, self.truncation = tokenizer, int(max_length), padding, truncation
        self._pin = bool(pin) and torch.cuda.is_available()
        self._rows = {}
        self._n = 0
        self._alloc(max(1, int(capacity)))
        self.hits = self.misses = 0

    def _alloc(self, cap: int) -> None:
        ids = torch.zeros(cap, self.max_length, dtype=torch.int64)
        mask = torch.zeros(cap, self.max_length, dtype=torch.int64)
        if self._pin:
            ids, mask = ids.pin_memory(), mask.pin_memory()
        if self._n:
            ids[:self._n].copy_(self.ids[:self._n])
            mask[:self._n].copy_(self.mask[:self._n])
        self.ids, self.mask = ids, mask

    @staticmethod
    def _key(caption: str, add_special_tokens: bool) -> str:
        import hashlib
        return hashlib.sha1((("1" if add_special_tokens else "0") + caption).encode("utf-8")).hexdigest()

    def row_of(self, caption: str, add_special_tokens: bool = True) -> int:
        key = self._key(caption, add_special_tokens)
        r = self._rows.get(key)
        if r is not None:
            self.hits += 1
            return r
        self.misses += 1
        try:
            enc = self.tokenizer(caption, max_length=self.max_length, padding=self.padding, truncation=self.truncation,
                                 add_special_tokens=add_special_tokens, return_tensors="pt", return_attention_mask=True)
        except Exception as e:  # reference preprocessing.py:236-238
            raise ValueError(f"Failed to encode caption: {e}") from e
        ids = torch.as_tensor(enc["input_ids"]).reshape(-1).to(torch.int64)
        am = enc.get("attention_mask") if hasattr(enc, "get") else None
        am = torch.ones_like(ids) if am is None else torch.as_tensor(am).reshape(-1).to(torch.int64)
        if ids.numel() > self.max_length:
            raise ValueError(f"Failed to encode caption: {ids.numel()} tokens exceed max_length={self.max_length} "
                             "(truncation is off)")
        if self._n == self.ids.shape[0]:
            self._alloc(2 * self._n)
        r = self._n
        n = ids.numel()
        pad = getattr(self.tokenizer, "pad_token_id", None)
        self.ids[r].fill_(0 if pad is None else int(pad))
        self.ids[r, :n] = ids
        self.mask[r, :n] = am
        self._rows[key] = r
        self._n += 1
        return r

    def encode_caption(self, caption: str, add_special_tokens: bool = True, return_attention_mask: bool = True):
        r = self.row_of(caption, add_special_tokens)
        return {"input_ids": self.ids[r].clone(), "attention_mask": self.mask[r].clone()}

    def encode_batch(self, captions, add_special_tokens: bool = True):
        """[B, max_length] ids and masks of a list of captions: one gather from the table."""
        rows = torch.as_tensor([self.row_of(c, add_special_tokens) for c in captions], dtype=torch.int64)
        return {"input_ids": self.ids.index_select(0, rows), "attention_mask": self.mask.index_select(0, rows)}

    def __len__(self) -> int:
        return self._n

    def save(self, path: str) -> None:
        import numpy as np
        keys = sorted(self._rows, key=self._rows.get)
        np.savez_compressed(path, ids=self.ids[:self._n].numpy(), mask=self.mask[:self._n].numpy(),
                            keys=np.array(keys), max_length=np.int64(self.max_length))

    def load(self, path: str) -> None:
        import numpy as np
        z = np.load(path, allow_pickle=False)
        if int(z["max_length"]) != self.max_length:
            raise ValueError(f"cache was built for max_length={int(z['max_length'])}, not {self.max_length}")
        n = z["ids"].shape[0]
        self._n = 0
        self._alloc(max(n, 1))
        self.ids[:n].copy_(torch.from_numpy(z["ids"]))
        self.mask[:n].copy_(torch.from_numpy(z["mask"]))
        self._rows = {str(k): i for i, k in enumerate(z["keys"])}
        self._n = n


def _pin(x):
    if isinstance(x, torch.Tensor) and not x.is_cuda and not x.is_pinned():
        return x.pin_memory()
    return x


def _tensors(obj):
    if isinstance(obj, torch.Tensor):
        yield obj
    elif isinstance(obj, dict):
        for v in obj.values():
            yield from _tensors(v)
    elif hasattr(obj, "__dataclass_fields__"):
        for k in obj.__dataclass_fields__:
            yield from _tensors(getattr(obj, k))


class BatchPrefetcher:
    """Iterates ``loader`` and yields ``prepare(batch, device)`` results whose device work was issued on a side stream
    up to ``depth`` batches ahead.  The consumer's current stream waits on the batch's event (no host block)."""

    def __init__(self, loader: Iterable, prepare: Callable, device, depth: int = 2):
        self.loader, self.prepare, self.device, self.depth = loader, prepare, torch.device(device), max(1, int(depth))

    def __len__(self) -> int:
        return len(self.loader)

    def __iter__(self) -> Iterator:
        q: "queue.Queue" = queue.Queue(maxsize=self.depth)
        side = torch.cuda.Stream(device=self.device)
        err = []

        def feed():
            try:
                torch.cuda.set_device(self.device)
                for batch in self.loader:
                    host = {k: _pin(v) for k, v in batch.items()} if isinstance(batch, dict) else batch
                    with torch.cuda.stream(side):
                        out = self.prepare(host, self.device)
                        ev = torch.cuda.Event()
                        ev.record(side)
                    q.put((out, ev, host))       # `host` keeps the pinned staging alive until the copy has run
            except BaseException as e:  # noqa: BLE001 - re-raised in the consumer
                err.append(e)
            finally:
                q.put(_STOP)

        th = threading.Thread(target=feed, name="pgca-prefetch", daemon=True)
        th.start()
        try:
            while True:
                item = q.get()
                if item is _STOP:
                    break
                out, ev, _host = item
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(ev)
                for t in _tensors(out):          # allocated on the side stream, used on this one
                    if t.is_cuda:
                        t.record_stream(cur)
                yield out
        finally:
            while th.is_alive():                 # drain so the feeder can finish if the consumer stopped early
                try:
                    q.get(timeout=0.1)
                except queue.Empty:
                    pass
            th.join()
        if err:
            raise err[0]
