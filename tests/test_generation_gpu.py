"""Caption generation on the HIP kernels (reference model.py:621-678, 855-923) against the oracle's
restatement of HF greedy decoding from ``inputs_embeds``; sampling / beam search: contract checks."""
import pytest
import torch

from oracle import restatement as R

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def model():
    from pgca_amd.arch import tiny_arch
    from pgca_amd.model import PreferenceGuidedCaptioningModel
    return PreferenceGuidedCaptioningModel(freeze_vision_backbone=True, arch=tiny_arch(), seed=17, device=DEV)


def test_greedy_matches_oracle(model):
    arch = model.arch
    g = torch.Generator().manual_seed(2)
    img = torch.randn(5, 3, arch.vit.image, arch.vit.image, generator=g)
    sd = {k: v.detach().cpu() for k, v in model.store.state_dict(aliases=False).items()}
    emb = R.vision_encoder_forward(sd, img, arch.vit.heads, arch.vit.patch)["embeddings"]
    pad, eos = arch.gpt.base_vocab, arch.gpt.base_vocab + 2
    want, margins = R.generate_greedy(sd, emb, 12, arch.gpt.heads, pad, eos, repetition_penalty=1.1)
    # first-step logits
    pv = model.caption_decoder.engine.prefix_embedding(model.vision_encoder(img)["embeddings"])
    l0 = model.caption_decoder.engine.next_token_logits(pv, torch.zeros(5, 0, dtype=torch.long, device=DEV)).cpu()
    ref0 = R.generate_step_logits(sd, emb, torch.zeros(5, 0, dtype=torch.long), arch.gpt.heads)
    assert float((l0 - ref0).abs().max()) <= 5e-2
    got = model.generate_token_ids(img, max_length=12, num_beams=1, do_sample=False, repetition_penalty=1.1).cpu()
    assert got.shape == want.shape
    for b in range(5):                      # identical until the first step whose top-2 margin is inside bf16 noise
        for t in range(want.shape[1]):
            if float(margins[b, t]) < 0.1:
                break
            assert int(got[b, t]) == int(want[b, t]), (b, t)


def test_sampling_and_beam_contracts(model):
    arch = model.arch
    img = torch.randn(3, 3, arch.vit.image, arch.vit.image, generator=torch.Generator().manual_seed(4))
    pad = arch.gpt.base_vocab
    gen = lambda s: torch.Generator(device=DEV).manual_seed(s)  # noqa: E731
    a = model.generate_token_ids(img, max_length=9, num_beams=1, do_sample=True, top_p=0.9, temperature=0.8, generator=gen(1))
    b = model.generate_token_ids(img, max_length=9, num_beams=1, do_sample=True, top_p=0.9, temperature=0.8, generator=gen(1))
    c = model.generate_token_ids(img, max_length=9, num_beams=1, do_sample=True, top_p=0.9, temperature=0.8, generator=gen(2))
    assert a.shape == (3, 9) and a.dtype == torch.int64 and torch.equal(a, b) and not torch.equal(a, c)
    assert int(a.max()) < arch.dec_vocab
    # beam search: never worse than greedy in total log-probability of the returned sequence
    beams = model.generate_token_ids(img, max_length=6, num_beams=4, do_sample=False, repetition_penalty=1.0)
    greedy = model.generate_token_ids(img, max_length=6, num_beams=1, do_sample=False, repetition_penalty=1.0)
    eng = model.caption_decoder.engine
    pv = eng.prefix_embedding(model.vision_encoder(img)["embeddings"])

    def seq_logp(ids):
        tot = torch.zeros(ids.shape[0], device=DEV)
        for t in range(ids.shape[1]):
            lp = torch.log_softmax(eng.next_token_logits(pv, ids[:, :t]), dim=-1)
            tot += lp.gather(1, ids[:, t:t + 1])[:, 0] * (ids[:, t] != pad)
        return tot
    assert bool((seq_logp(beams) >= seq_logp(greedy) - 1e-3).all())
    # eos stops a sequence and pads the rest
    first = int(greedy[0, 0])
    stopped = model.generate_token_ids(img[:1], max_length=6, num_beams=1, do_sample=False, repetition_penalty=1.0,
                                       eos_token_id=first)
    assert stopped.shape[1] == 1 and int(stopped[0, 0]) == first
    with pytest.raises(RuntimeError, match="tokenizer"):
        model.generate_captions(img)

    class Tok:
        def decode(self, ids, skip_special_tokens=True):
            return " ".join(str(i) for i in ids if i < pad)
    model.caption_decoder.tokenizer = Tok()
    caps = model.generate_captions(img, max_length=4, num_beams=1, do_sample=False)
    assert len(caps) == 3 and all(isinstance(c, str) and c for c in caps)
    del model.caption_decoder.tokenizer
