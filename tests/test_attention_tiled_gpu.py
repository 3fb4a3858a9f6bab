"""Key-tiled attention (csrc/attention_tiled.hip) against fp32 PyTorch: the sequence lengths of configs C4/C5
(GPT-2 captions at S = 256: causal + right padding + replayed probability dropout; ViT-L/14 at T = 257: no mask),
and ragged lengths around the 128-row block edges (S <= 128 is covered by tests/test_kernels_gpu.py)."""
import pytest
import torch

from test_kernels_gpu import attn_ref, close, dev, drop_mult, hip, rnd  # noqa: F401  (hip is a fixture)

pytestmark = pytest.mark.gpu


def _lens(B, S):
    return torch.tensor([S, max(1, S // 2 + 3), max(1, S - 1), 1, 129, 17][:B]).clamp(max=S)


@pytest.mark.parametrize("cfg", [(3, 256, 2, True), (2, 257, 3, False), (4, 129, 1, True), (2, 200, 2, True),
                                 (2, 384, 1, True), (3, 300, 2, False), (2, 512, 1, True), (6, 256, 1, True)])
def test_tiled_attention_fwd_bwd(hip, cfg):
    B, S, heads, causal = cfg
    H = heads * 64
    qkv = rnd(B * S, 3 * H, seed=S + B).bfloat16()
    mask = (torch.arange(S)[None] < _lens(B, S)[:, None]).int().to(dev())
    if not causal:
        mask = torch.ones_like(mask)
    out = torch.full((B * S, H), 7.0, dtype=torch.bfloat16, device=dev())
    lse = torch.zeros(B, heads, S, device=dev())
    hip.attention_fwd(qkv, mask, B, S, heads, causal, out, lse)
    x = qkv.float().requires_grad_()
    ref, ref_lse = attn_ref(x, mask, B, S, heads, causal)
    close(out, ref, 1.0 / 64, "attn out")
    assert float((lse - ref_lse).abs().max()) <= 2e-3
    valid = (mask.view(B * S, 1) != 0).float()
    dout = (rnd(B * S, H, seed=7) * valid).bfloat16()
    ref.backward(dout.float())
    dqkv = torch.full((B * S, 3 * H), 7.0, dtype=torch.bfloat16, device=dev())
    hip.attention_bwd(qkv, out, dout, lse, mask, B, S, heads, causal, dqkv)
    close(dqkv, x.grad, 1.0 / 40, "attn dqkv")
    for name, sl in (("dq", slice(0, H)), ("dk", slice(H, 2 * H)), ("dv", slice(2 * H, 3 * H))):
        a, b = dqkv[:, sl].float().flatten().double(), x.grad[:, sl].flatten().double()
        assert float(a @ b / (a.norm() * b.norm())) >= 0.9995, name
    pad_rows = (mask.view(-1) == 0)
    if bool(pad_rows.any()):   # padded keys receive exactly zero gradient
        assert float(dqkv[pad_rows][:, H:].abs().max()) == 0.0


@pytest.mark.parametrize("S", [256, 160])
def test_tiled_attention_dropout_replay(hip, S):
    B, heads = 3, 2
    H = heads * 64
    qkv = rnd(B * S, 3 * H, seed=11).bfloat16()
    mask = (torch.arange(S)[None] < torch.tensor([S, 150, 16])[:, None]).int().to(dev())
    d = hip.drop_args(12345, 0.1)
    out = torch.zeros(B * S, H, dtype=torch.bfloat16, device=dev())
    lse = torch.zeros(B, heads, S, device=dev())
    hip.attention_fwd(qkv, mask, B, S, heads, True, out, lse, drop=d)
    mult = drop_mult(12345, 0.1, (B, heads, S, S))
    x = qkv.float().requires_grad_()
    q, k, v = x.view(B, S, 3, heads, 64).permute(2, 0, 3, 1, 4)
    s = (q @ k.transpose(-1, -2)) * 0.125
    allowed = torch.tril(torch.ones(S, S, dtype=torch.bool, device=dev()))[None, None] & (mask[:, None, None, :] != 0)
    p = torch.softmax(s.masked_fill(~allowed, float("-inf")), -1) * mult
    ref = (p @ v).permute(0, 2, 1, 3).reshape(B * S, H)
    close(out, ref, 1.0 / 64, "attn + prob dropout")
    valid = (mask.view(B * S, 1) != 0).float()
    dout = (rnd(B * S, H, seed=7) * valid).bfloat16()
    ref.backward(dout.float())
    dqkv = torch.zeros(B * S, 3 * H, dtype=torch.bfloat16, device=dev())
    hip.attention_bwd(qkv, out, dout, lse, mask, B, S, heads, True, dqkv, drop=d)
    close(dqkv, x.grad, 1.0 / 40, "attn bwd with replayed dropout")


def test_tiled_attention_is_reproducible(hip):
    """No atomics anywhere: two launches on the same inputs are bitwise equal."""
    B, S, heads = 4, 256, 4
    H = heads * 64
    qkv = rnd(B * S, 3 * H, seed=5).bfloat16()
    mask = (torch.arange(S)[None] < _lens(B, S)[:, None]).int().to(dev())
    res = []
    for _ in range(2):
        out = torch.zeros(B * S, H, dtype=torch.bfloat16, device=dev())
        lse = torch.zeros(B, heads, S, device=dev())
        dqkv = torch.zeros(B * S, 3 * H, dtype=torch.bfloat16, device=dev())
        hip.attention_fwd(qkv, mask, B, S, heads, True, out, lse)
        hip.attention_bwd(qkv, out, rnd(B * S, H, seed=9).bfloat16(), lse, mask, B, S, heads, True, dqkv)
        res.append((out.clone(), lse.clone(), dqkv.clone()))
    for a, b in zip(*res):
        assert torch.equal(a, b)
