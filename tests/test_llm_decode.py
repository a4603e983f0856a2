"""Graphed LLM decode: token parity with HF generate."""
import pytest
import torch

from llamole_amd import e2e
from llamole_amd.llm_decode import GraphedDecoder, sample_top_p


def _case(device, dtype):
    llm = e2e.build_llm("tiny", device, dtype)
    g = torch.Generator().manual_seed(0)
    prompt = torch.randint(5, 1000, (2, 12), generator=g).to(device)
    mask = torch.ones_like(prompt)
    mask[1, :4] = 0
    return llm, prompt, mask


def test_eager_static_cache_equals_hf_generate_cpu():
    llm, prompt, mask = _case("cpu", torch.float32)
    ref = llm.generate(inputs=prompt, attention_mask=mask, max_new_tokens=10, do_sample=False, pad_token_id=0, eos_token_id=[2047])
    got = GraphedDecoder(llm, use_graph=False).generate(prompt, mask, max_new_tokens=10, do_sample=False, pad_token_id=0,
                                                        eos_token_id=[2047])
    assert torch.equal(ref, got)
    # stopping: make the first sampled token an EOS for row 0 -> the rest of the row is padding
    first = int(ref[0, 12])
    got2 = GraphedDecoder(llm, use_graph=False, sync_every=1).generate(prompt, mask, max_new_tokens=10, do_sample=False,
                                                                      pad_token_id=0, eos_token_id=[first])
    assert int(got2[0, 12]) == first and (got2[0, 13:] == 0).all()


def test_top_p_sampler_support():
    torch.manual_seed(0)
    logits = torch.tensor([[4.0, 3.0, 0.0, -2.0, -9.0]]).repeat(2000, 1)
    s = sample_top_p(logits, temperature=0.6, top_p=0.9)
    assert set(s.tolist()) <= {0, 1}          # tokens outside the 0.9 nucleus are never drawn
    assert 0.75 < (s == 0).float().mean() < 0.92


@pytest.mark.gpu
def test_graph_replay_equals_hf_generate_gpu():
    llm, prompt, mask = _case("cuda", torch.float32)
    ref = llm.generate(inputs=prompt, attention_mask=mask, max_new_tokens=24, do_sample=False, pad_token_id=0, eos_token_id=[2047])
    dec = GraphedDecoder(llm, use_graph=True)
    got = dec.generate(prompt, mask, max_new_tokens=24, do_sample=False, pad_token_id=0, eos_token_id=[2047])
    assert torch.equal(ref, got)
    got2 = dec.generate(prompt, mask, max_new_tokens=24, do_sample=False, pad_token_id=0, eos_token_id=[2047])   # graph reuse
    assert torch.equal(ref, got2)
    emb = llm.get_input_embeddings()(prompt)
    ref3 = llm.generate(inputs_embeds=emb, attention_mask=mask, max_new_tokens=24, do_sample=False, pad_token_id=0, eos_token_id=[2047])
    got3 = GraphedDecoder(llm, use_graph=True).generate(None, mask, inputs_embeds=emb, max_new_tokens=24, do_sample=False,
                                                        pad_token_id=0, eos_token_id=[2047])
    assert torch.equal(ref3, got3)
