"""ll_decode_chain_bf16 (o_proj + residual -> RMSNorm + gate|up + SiLU*mul -> down_proj + residual -> the next layer's RMSNorm + q|k|v as
ONE launch with phase counters) against the four ll_gemv_fused_bf16 launches it replaces: bit-identical by construction (same FMA chains,
same reductions, same roundings) -- at the C ABI over decode shapes, launched back to back many times (the counters must come back to
zero every time), and through the decoder (LLAMOLE_DECODE_CHAIN on / off: same tokens, logits and KV cache)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


SHAPES = {      # H, nq, I, nqkv_next
    "qwen2-7b": (3584, 3584, 18944, 4608),
    "llama-3.1-8b": (4096, 4096, 14336, 6144),
    "tiny": (256, 256, 512, 384),
    "ragged": (200, 264, 328, 136),          # multiples of 8 only: partial workgroups in every phase
    "last_layer": (512, 512, 1024, 0),       # no next layer: three phases
}


def _inputs(H, nq, I, nd, seed):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc)      # noqa: E731
    t = dict(att=r(1, nq).bfloat16(), wo=r(H, nq, sc=nq ** -0.5).bfloat16(), bo=r(H, sc=0.1).float(), res=r(1, H).bfloat16(),
             wgu=r(2 * I, H, sc=H ** -0.5).bfloat16(), norm2=(1 + 0.1 * r(H)).bfloat16(), wdown=r(H, I, sc=I ** -0.5).bfloat16(),
             wqkv=r(max(nd, 8), H, sc=H ** -0.5).bfloat16(), bqkv=r(max(nd, 8), sc=0.1).float(), norm1=(1 + 0.1 * r(H)).bfloat16())
    return {k: v.cuda() for k, v in t.items()}


def _four_launches(lib, t, H, nq, I, nd, use_bias, outs=None):
    s = torch.cuda.current_stream().cuda_stream
    from llamole_amd import _lib
    mk = lambda n: torch.full((1, n), float("nan"), dtype=torch.bfloat16, device="cuda")      # noqa: E731
    h1, act, h2, qkv = outs or (mk(H), mk(I), mk(H), mk(max(nd, 8)))
    bo = t["bo"].data_ptr() if use_bias else None
    _lib.check(lib.ll_gemv_fused_bf16(t["att"].data_ptr(), nq, t["wo"].data_ptr(), nq, bo, None, 0.0, t["res"].data_ptr(), H, h1.data_ptr(), H,
                                      1, H, nq, 1, s), "o_proj")
    _lib.check(lib.ll_gemv_fused_bf16(h1.data_ptr(), H, t["wgu"].data_ptr(), H, None, t["norm2"].data_ptr(), 1e-6, None, 0, act.data_ptr(), I,
                                      1, I, H, 2, s), "gate|up")
    _lib.check(lib.ll_gemv_fused_bf16(act.data_ptr(), I, t["wdown"].data_ptr(), I, None, None, 0.0, h1.data_ptr(), H, h2.data_ptr(), H,
                                      1, H, I, 1, s), "down_proj")
    if nd:
        _lib.check(lib.ll_gemv_fused_bf16(h2.data_ptr(), H, t["wqkv"].data_ptr(), H, t["bqkv"].data_ptr() if use_bias else None,
                                          t["norm1"].data_ptr(), 1e-5, None, 0, qkv.data_ptr(), nd, 1, nd, H, 0, s), "q|k|v")
    return h1, act, h2, qkv


def _chain(lib, t, H, nq, I, nd, use_bias, ctr, outs=None):
    from llamole_amd import _lib
    mk = lambda n: torch.full((1, n), float("nan"), dtype=torch.bfloat16, device="cuda")      # noqa: E731
    h1, act, h2, qkv = outs or (mk(H), mk(I), mk(H), mk(max(nd, 8)))
    _lib.check(lib.ll_decode_chain_bf16(t["att"].data_ptr(), t["wo"].data_ptr(), t["bo"].data_ptr() if use_bias else None, t["res"].data_ptr(),
                                        h1.data_ptr(), H, nq, t["wgu"].data_ptr(), t["norm2"].data_ptr(), 1e-6, act.data_ptr(), I,
                                        t["wdown"].data_ptr(), h2.data_ptr(), t["wqkv"].data_ptr() if nd else None,
                                        (t["bqkv"].data_ptr() if use_bias else None) if nd else None, t["norm1"].data_ptr() if nd else None, 1e-5,
                                        qkv.data_ptr() if nd else None, nd, ctr.data_ptr(), torch.cuda.current_stream().cuda_stream),
               "ll_decode_chain_bf16")
    return h1, act, h2, qkv


def _error(lib, ctr):
    from llamole_amd import _lib
    e = C.c_uint(7)
    _lib.check(lib.ll_decode_chain_error(ctr.data_ptr(), C.byref(e)), "ll_decode_chain_error")
    return int(e.value)


@pytest.mark.parametrize("name", list(SHAPES))
@pytest.mark.parametrize("use_bias", [False, True])
def test_chain_equals_four_gemv_launches(name, use_bias):
    from llamole_amd import _lib
    lib = _lib.load()
    H, nq, I, nd = SHAPES[name]
    t = _inputs(H, nq, I, nd, seed=len(name))
    want = _four_launches(lib, t, H, nq, I, nd, use_bias)
    ctr = torch.zeros(8192, dtype=torch.int32, device="cuda")
    got = _chain(lib, t, H, nq, I, nd, use_bias, ctr)
    torch.cuda.synchronize()
    for w, g_, what in zip(want, got, ("h1", "act", "h2", "qkv_next")):
        if what == "qkv_next" and not nd:
            continue
        n = nd if what == "qkv_next" else w.shape[1]
        assert torch.equal(w[:, :n], g_[:, :n]), (name, what, float((w[:, :n].float() - g_[:, :n].float()).abs().max()))
        assert torch.isfinite(g_[:, :n].float()).all()
    assert _error(lib, ctr) == 0
    assert int(ctr.abs().sum()) == 0, "the launch leaves its counters zeroed"


def test_chain_back_to_back_launches_reuse_the_counters():
    """A decode loop launches the chain of every layer once per token on the same buffers and counters: 300 launches in a row with a new
    input each (the previous launch's h2 as the next residual), compared with the four-launch evaluation of the same recurrence."""
    from llamole_amd import _lib
    lib = _lib.load()
    H, nq, I, nd = SHAPES["qwen2-7b"]
    t = _inputs(H, nq, I, nd, seed=11)
    ctr = torch.zeros(8192, dtype=torch.int32, device="cuda")
    mk = lambda n: torch.zeros(1, n, dtype=torch.bfloat16, device="cuda")      # noqa: E731
    outs = (mk(H), mk(I), mk(H), mk(nd))
    res0 = t["res"].clone()
    chain_rows = []
    for i in range(300):
        h1, act, h2, qkv = _chain(lib, t, H, nq, I, nd, True, ctr, outs)
        t["att"] = qkv[:, :nq].clone()                 # next token's "attention output": whatever the chain left, so errors propagate
        t["res"] = (h2 * 0.5).clone()
        if i % 50 == 49:
            chain_rows.append((h2.clone(), qkv.clone()))
    torch.cuda.synchronize()
    assert _error(lib, ctr) == 0 and int(ctr.abs().sum()) == 0
    t["att"], t["res"] = _inputs(H, nq, I, nd, seed=11)["att"], res0
    k = 0
    for i in range(300):
        h1, act, h2, qkv = _four_launches(lib, t, H, nq, I, nd, True)
        t["att"] = qkv[:, :nq].clone()
        t["res"] = (h2 * 0.5).clone()
        if i % 50 == 49:
            assert torch.equal(h2, chain_rows[k][0]) and torch.equal(qkv, chain_rows[k][1]), i
            k += 1


def test_chain_refuses_shapes_outside_its_envelope():
    from llamole_amd import _lib
    lib = _lib.load()
    H, nq, I, nd = SHAPES["tiny"]
    t = _inputs(H, nq, I, nd, seed=1)
    ctr = torch.zeros(8192, dtype=torch.int32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    buf = torch.zeros(1, 32768, dtype=torch.bfloat16, device="cuda")

    def call(H_, nq_, I_, nd_, counters=ctr.data_ptr()):
        return lib.ll_decode_chain_bf16(t["att"].data_ptr(), t["wo"].data_ptr(), None, t["res"].data_ptr(), buf.data_ptr(), H_, nq_,
                                        t["wgu"].data_ptr(), t["norm2"].data_ptr(), 1e-6, buf.data_ptr(), I_, t["wdown"].data_ptr(),
                                        buf.data_ptr(), t["wqkv"].data_ptr(), None, t["norm1"].data_ptr(), 1e-5, buf.data_ptr(), nd_,
                                        counters, s)
    assert call(H + 4, nq, I, nd) == -1            # LL_EINVAL: not a multiple of 8
    assert call(H, nq, I + 2, nd) == -1
    assert call(8200, nq, I, nd) == -1             # beyond the LDS row
    assert call(H, nq, 20488, nd) == -1
    assert call(H, nq, I, nd, counters=None) == -1
    torch.cuda.synchronize()
    assert _error(lib, ctr) == 0


@pytest.mark.parametrize("arch", ["tiny", "tiny-llama", "tiny-mistral"])
def test_decoder_with_chain_equals_five_launch_layers(arch, monkeypatch):
    """The decoder with the chain (two launches per layer: attention, chain) against the five-launch layers and against the one-launch-per-op
    accelerated path: tokens, logits and the whole KV cache, eager and as a hipGraph replayed over different prompts; 40 new tokens."""
    from llamole_amd import e2e, llm_accel
    from llamole_amd.llm_decode import GraphedDecoder
    llm = e2e.build_llm(arch, "cuda", torch.bfloat16)
    prompt = torch.randint(5, 1000, (1, 13), generator=torch.Generator().manual_seed(4)).cuda()
    mask = torch.ones_like(prompt)
    kw = dict(max_new_tokens=40, do_sample=False, pad_token_id=0, eos_token_id=[])
    try:
        llm_accel.accelerate_linears(llm, min_weight_elems=1)
        llm_accel.accelerate_elementwise(llm)
        assert llm_accel.use_decode_attention(llm)
        base = GraphedDecoder(llm, use_graph=False, fused_cache=True)
        ref = base.generate(prompt, mask, **kw)
        ref_logits = base.last_logits.clone()
        ref_kv = [(l.keys.clone(), l.values.clone()) for l in base.cache.layers]
        n = llm_accel.fuse_decoder_layers(llm)
        assert n == llm.config.num_hidden_layers and llm_accel.fuse_model_decode(llm)
        fused = [l._ll_fused for l in llm.model.layers]
        assert all(f.chain_ok for f in fused) and all(a.next is b for a, b in zip(fused, fused[1:])) and fused[-1].next is None
        results = {}
        for chain in (False, True):
            monkeypatch.setattr(llm_accel, "DECODE_CHAIN", chain)
            for use_graph in (False, True):
                d = GraphedDecoder(llm, use_graph=use_graph, fused_cache=True)
                got = d.generate(prompt, mask, **kw)
                assert torch.equal(got, ref), (chain, use_graph)
                assert torch.equal(d.last_logits, ref_logits), (chain, use_graph)
                for (k, v), l in zip(ref_kv, d.cache.layers):
                    assert torch.equal(l.keys, k) and torch.equal(l.values, v)
                if use_graph:          # the captured graph again, on another prompt of another length, then the first one again
                    p2 = torch.randint(5, 1000, (1, 7), generator=torch.Generator().manual_seed(9)).cuda()
                    results[chain] = d.generate(p2, torch.ones_like(p2), **kw)
                    assert torch.equal(d.generate(prompt, mask, **kw), ref)
            used = [f._chain_bufs is not None for f in fused]
            assert all(used) if chain else True
        assert torch.equal(results[False], results[True])
        assert all(f.chain_error() == 0 for f in fused)
        # a layer HF calls on its own (hidden-state taps, hooks) keeps fresh outputs: five launches, no carried q|k|v
        monkeypatch.setattr(llm_accel, "DECODE_CHAIN", True)
        llm_accel.restore_model_decode(llm)
        d = GraphedDecoder(llm, use_graph=False, fused_cache=True)
        assert torch.equal(d.generate(prompt, mask, **kw), ref) and all(f._carry is None for f in fused)
    finally:
        llm_accel.restore_model_decode(llm)
        llm_accel.restore_decoder_layers(llm)
        llm_accel.restore_elementwise(llm)
        llm_accel.restore_linears(llm)
