"""CPU tests of the drop-in wrapper classes: file formats, state-dict keys, table construction, error behaviour.
(The arithmetic needs a HIP device and is covered by the -m gpu tests; here nothing is computed on the engine.)"""
import json
import os

import numpy as np
import pytest
import torch

from llamole_amd import synth
from tests.cases import DIT_CASES, GIN_CASES, dit_case, load_golden


@pytest.fixture()
def dit_dir(tmp_path):
    cfg, meta, sd, B, seed = dit_case("dit_n32_h128")
    synth.write_dit_dir(str(tmp_path), cfg, meta, sd)
    return str(tmp_path), cfg, meta, sd


def test_graphdit_tables_match_reference_goldens(dit_dir):
    from llamole_amd.graph_decoder import GraphDiT
    d, cfg, meta, sd = dit_dir
    m = GraphDiT(os.path.join(d, "config.yaml"), os.path.join(d, "data.meta.json"), torch.float32)
    g = load_golden("dit_n32_h128")
    np.testing.assert_allclose(m.tables["betas"].numpy(), g["betas"], rtol=1e-6)
    np.testing.assert_allclose(m.tables["alphas_bar"].numpy(), g["alphas_bar"], rtol=1e-6)
    np.testing.assert_allclose(m.tables["x_marg"].numpy(), g["x_marg"], rtol=1e-6)
    np.testing.assert_allclose(m.tables["e_marg"].numpy(), g["e_marg"], rtol=1e-6)
    N = m.max_n_nodes
    # u = [[u_x, tile(u_xe)], [tile(u_ex), tile(u_e)]]  (reference diffusion_utils.py:296-305)
    np.testing.assert_allclose(m.tables["u_xe"].numpy(), g["u"][:16, 16:21], rtol=1e-6)
    np.testing.assert_allclose(m.tables["u_ex"].numpy(), g["u"][16:21, :16], rtol=1e-6)
    assert m.T == cfg["diffusion_steps"] and m.guide_scale == cfg["guide_scale"] and m.hidden_size == cfg["hidden_size"]
    assert m.text_input_size == 768 and m.atom_decoder == meta["active_atoms"] and N == meta["max_node"]


def test_graphdit_state_dict_roundtrip_and_errors(dit_dir, tmp_path):
    from llamole_amd.graph_decoder import GraphDiT
    d, cfg, meta, sd = dit_dir
    m = GraphDiT(os.path.join(d, "config.yaml"), os.path.join(d, "data.meta.json"), torch.float32)
    assert list(m.denoiser.state_dict().keys()) == list(sd.keys())          # reference Transformer key order
    m.init_model(d)
    for k, v in m.denoiser.state_dict().items():
        assert torch.equal(v, sd[k]), k
    out = tmp_path / "saved"
    m.save_pretrained(str(out))
    assert sorted(os.listdir(out)) == ["data.meta.json", "model.pt", "model_config.yaml"]
    assert json.load(open(out / "data.meta.json"))["max_node"] == meta["max_node"]
    m.disable_grads()
    assert not any(p.requires_grad for p in m.parameters())
    with pytest.raises(FileNotFoundError):
        m.init_model(str(tmp_path / "missing"))
    with pytest.raises(FileNotFoundError):
        GraphDiT(os.path.join(d, "nope.yaml"), os.path.join(d, "data.meta.json"), torch.float32)
    with pytest.raises(FileNotFoundError):
        GraphDiT(os.path.join(d, "config.yaml"), os.path.join(d, "nope.json"), torch.float32)
    props, text, n_nodes = synth.make_dit_inputs(2, 0, meta["max_node"])
    with pytest.raises(RuntimeError, match="HIP device"):      # no CPU path in the product
        m.generate_graphs(props, text, -200.0)
    with pytest.raises(RuntimeError, match="HIP device"):      # the training forward has no CPU path either
        m(torch.zeros(2, dtype=torch.long), torch.zeros((2, 0), dtype=torch.long), torch.zeros(0, dtype=torch.long),
          torch.zeros(2, dtype=torch.long), props[:1], text[:1], -200.0)
    torch.manual_seed(0)
    n = m.sample_n_nodes(1000)
    assert int(n.min()) >= 5 and int(n.max()) <= meta["max_node"]


def test_gin_wrappers_keys_files_and_errors(tmp_path):
    from llamole_amd.graph_encoder import GraphCLIP
    from llamole_amd.graph_predictor import GraphPredictor
    L, H, out_dim, G, seed = GIN_CASES["gin_l3_h64"]
    enc = GraphCLIP(L, H, 0.0, {"num_layer": L, "hidden_size": H, "drop_ratio": 0.0})
    assert sorted(enc.molecule_encoder.state_dict()) == sorted(synth.gin_weight_shapes(L, H, "encoder"))
    assert sorted(enc.molecule_projection.state_dict()) == sorted(synth.proj_weight_shapes(H))
    enc.molecule_encoder.load_state_dict(synth.make_gin_weights(L, H, "encoder", seed=seed))
    enc.save_pretrained(str(tmp_path / "enc"))
    assert sorted(os.listdir(tmp_path / "enc")) == ["model.pt", "model_config.json", "model_proj.pt"]
    enc2 = GraphCLIP(L, H, 0.0, {})
    enc2.init_model(str(tmp_path / "enc"), verbose=False)
    assert torch.equal(enc2.molecule_encoder.state_dict()["convs.0.eps"], enc.molecule_encoder.state_dict()["convs.0.eps"])
    with pytest.raises(FileNotFoundError):
        enc2.init_model(str(tmp_path / "none"))
    with pytest.raises(ValueError, match="greater than 1"):
        GraphCLIP(1, H, 0.0, {})
    x, ei, ea, batch = synth.make_mol_graphs(2, 0)
    with pytest.raises(RuntimeError, match="HIP device"):
        enc(x, ei, ea, batch)

    import pandas as pd
    l2t = pd.DataFrame({"rule_label": [0, 1, 2], "retro_templates": ["a>>b", "c>>d", "e>>f"]})
    pred = GraphPredictor(L, H, 0.1, 3, {"num_layer": L, "hidden_size": H, "drop_ratio": 0.1, "num_task": 3}, l2t, available=["CCO", "CC"])
    assert pred.label_to_template == {0: "a>>b", 1: "c>>d", 2: "e>>f"} and pred.text_input_size == 768
    assert sorted(pred.predictor.state_dict()) == sorted(synth.gin_weight_shapes(L, H, "predictor", 3))
    torch.save(synth.make_cost_weights(0), tmp_path / "cost_model.pt")
    pred.init_neural_cost(str(tmp_path))
    pred.save_pretrained(str(tmp_path / "pred"))
    assert sorted(os.listdir(tmp_path / "pred")) == ["available.csv.gz", "cost_model.pt", "label_to_template.csv.gz", "model.pt", "model_config.json"]
    back = pd.read_csv(tmp_path / "pred" / "label_to_template.csv.gz", compression="gzip")
    assert list(back.columns) == ["rule_label", "retro_templates"] and len(back) == 3
    with pytest.raises(ValueError, match="not initialized"):
        GraphPredictor(L, H, 0.1, 3, {}, {}).estimate_cost("CCO")


def test_loader_seam(tmp_path):
    """load_graph_* keep the reference signatures / layouts; missing files -> FileNotFoundError (no network here)."""
    import types
    from llamole_amd import loader
    cfg, meta, sd, B, seed = dit_case("dit_n32_h128")
    d = tmp_path / "dit"
    synth.write_dit_dir(str(d), cfg, meta, sd)
    args = types.SimpleNamespace(compute_dtype=torch.bfloat16, disable_graph_model_gradient=True)
    m = loader.load_graph_decoder(args, str(d), "cpu")
    assert all(p.dtype == torch.bfloat16 and not p.requires_grad for p in m.parameters())   # reference loader.py:241-247
    with pytest.raises(FileNotFoundError):
        loader.load_graph_encoder(args, str(tmp_path / "no_enc"), "cpu")
    with pytest.raises(FileNotFoundError):
        loader.load_graph_predictor(args, str(tmp_path / "no_pred"), "cpu")
    L, H = 3, 64
    e = tmp_path / "enc"
    os.makedirs(e)
    json.dump({"num_layer": L, "hidden_size": H, "drop_ratio": 0.0}, open(e / "config.json", "w"))
    torch.save(synth.make_gin_weights(L, H, "encoder"), e / "model.pt")
    torch.save(synth.make_proj_weights(H), e / "model_proj.pt")
    enc = loader.load_graph_encoder(args, str(e), "cpu")
    assert enc.hidden_size == H and next(enc.parameters()).dtype == torch.bfloat16


def test_graph_batch_roundtrip():
    from llamole_amd.graph_data import GraphBatch, GraphData
    x, ei, ea, batch = synth.make_mol_graphs(3, 1)
    sizes = torch.bincount(batch).tolist()
    parts, off = [], 0
    for n in sizes:
        sel = (ei[0] >= off) & (ei[0] < off + n)
        parts.append(GraphData(x[off:off + n], ei[:, sel] - off, ea[sel]))
        off += n
    gb = GraphBatch.from_data_list(parts)
    assert torch.equal(gb.x, x) and torch.equal(gb.batch, batch) and torch.equal(gb.edge_index, ei) and torch.equal(gb.edge_attr, ea)
    back = gb.to_data_list()
    assert [d.num_nodes for d in back] == sizes and torch.equal(back[1].edge_index, parts[1].edge_index)


def test_merge_lora_adapter_with_modules_to_save(tmp_path):
    """reference adapters trained with resize_vocab carry embed_tokens / lm_head as modules_to_save (reference
    adapter.py:224-233); peft stores them as plain `<module>.weight`.  They must be copied (not crash on `tensor or tensor`),
    and a shape mismatch (embeddings not resized) must raise instead of being skipped."""
    import pytest
    from safetensors.torch import load_file
    from llamole_amd import e2e, synth
    from llamole_amd.sft import merge_lora_adapter
    llm = e2e.build_llm("tiny", "cpu", torch.float32)
    synth.write_lora_adapter_dir(str(tmp_path / "ad"), llm)
    t = load_file(str(tmp_path / "ad" / "adapter_model.safetensors"))
    assert "base_model.model.lm_head.weight" in t and "base_model.model.model.embed_tokens.weight" in t
    w0 = llm.model.layers[0].self_attn.q_proj.weight.detach().clone()
    assert merge_lora_adapter(llm, str(tmp_path / "ad")) == 6
    torch.testing.assert_close(llm.lm_head.weight.detach(), t["base_model.model.lm_head.weight"])
    torch.testing.assert_close(llm.model.embed_tokens.weight.detach(), t["base_model.model.model.embed_tokens.weight"])
    a = t["base_model.model.model.layers.0.self_attn.q_proj.lora_A.weight"]
    b = t["base_model.model.model.layers.0.self_attn.q_proj.lora_B.weight"]
    torch.testing.assert_close(llm.model.layers[0].self_attn.q_proj.weight.detach(), w0 + 2.0 * (b @ a), rtol=1e-5, atol=1e-6)
    small = e2e.build_llm("tiny", "cpu", torch.float32)
    small.resize_token_embeddings(small.config.vocab_size - 64)          # the base model was NOT resized to the adapter's vocabulary
    with pytest.raises(ValueError, match="resize"):
        merge_lora_adapter(small, str(tmp_path / "ad"))


def test_csr_error_names_the_call_that_converted_the_batch(monkeypatch):
    """a malformed batch is reported by a LATER call; the message must name the call that handed the batch over, and a flag
    that has already arrived must not cost a device synchronisation."""
    from llamole_amd import graph_encoder as ge
    ring = torch.zeros(256, dtype=torch.int32)
    monkeypatch.setattr(ge, "_flag_ring", ring)
    monkeypatch.setattr(ge, "_pending_csr_flags", [])
    monkeypatch.setattr(ge, "_flag_origin", {})
    syncs = []
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: syncs.append(a))
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)

    def convert():                       # stands in for graph_csr_device: takes a slot, the "kernel" answers later
        slot = ge._flag_slot()
        ge._pending_csr_flags.append(slot)
        return slot
    s0 = convert()
    ring[s0] = 0                         # fine batch, flag arrived
    ge.check_graph_errors(wait=True)
    assert not syncs and not ge._pending_csr_flags
    s1 = convert()
    ge.check_graph_errors()              # not arrived yet, no wait: nothing happens
    assert ge._pending_csr_flags == [s1] and not syncs
    ring[s1] = 2                         # the kernel found an edge outside the batch
    with pytest.raises(ValueError, match=r"outside the batch.*\[batch converted at test_wrappers_cpu.py:\d+ \(convert\)\]"):
        ge.check_graph_errors(wait=True)
    assert not syncs                     # the flag was there: polled, not waited for
    s2 = convert()                       # still -1 after the poll: only now the devices are synchronised
    ge.check_graph_errors(wait=True)
    assert len(syncs) == 1 and ge._pending_csr_flags == [s2]


def test_host_timeline_marks_are_off_unless_started():
    from llamole_amd import _trace
    _trace.mark("ignored")
    assert _trace.events is None
    _trace.start()
    _trace.mark("a")
    _trace.mark("b")
    ev = _trace.stop()
    assert [n for n, _ in ev] == ["a", "b"] and ev[0][1] <= ev[1][1] and _trace.events is None
