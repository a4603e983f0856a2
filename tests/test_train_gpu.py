"""`python main.py train cfg.yaml` end to end on the GPU box (reference src/train/mmsft/workflow.py:41-118,
modeling_llamole.py:439-519, trainer.py:232-234): LoRA SFT of a tiny HF language model with the graph side of the loss on the HIP
engines, the checkpoint the reference layout prescribes, `main.py eval` loading that checkpoint back, resuming, and two ranks."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fake_chem(monkeypatch):
    from llamole_amd.graph_data import GraphData
    from llamole_amd.modeling_llamole import GraphLLMForCausalMLM
    from tests.host_fakes import fake_smiles_to_graph
    to_graph = fake_smiles_to_graph(GraphData)
    monkeypatch.setattr(GraphLLMForCausalMLM, "smiles_to_graph", lambda self, s: to_graph(s))


def test_main_train_saves_what_it_trains_and_eval_loads_it(tmp_path, monkeypatch):
    import yaml
    from llamole_amd import synth
    from llamole_amd import train as tr
    from llamole_amd.modeling_llamole import SPECIAL_TOKENS
    _fake_chem(monkeypatch)
    cfg = synth.write_train_fixture(str(tmp_path), SPECIAL_TOKENS, num_train_epochs=10.0)
    torch.manual_seed(0)
    out = tr.run_train(cfg)
    losses = [r["loss"] for r in out["log"]]
    assert len(losses) == 10 and all(l == l for l in losses)
    assert min(losses[-3:]) < losses[0], losses                       # it learns something on six records
    assert all(r["retro_loss"] > 0 and r["lm_loss"] > 0 for r in out["log"])
    od = out["output_dir"]
    for f in ("adapter_model.safetensors", "adapter_config.json", "graphllm_config.json", "connector/graph_to_lm_connector.pt",
              "connector/lm_to_graph_decoder.pt", "connector/lm_to_graph_predictor.pt", "trainer_log.jsonl", "train_results.json"):
        assert os.path.exists(os.path.join(od, f)), f
    gcfg = json.load(open(os.path.join(od, "graphllm_config.json")))
    assert gcfg["num_body_tokens"] == 8 and "<retro_start>" in gcfg["token_id_dict"] and gcfg["loss_weight_retro"] == 1
    acfg = json.load(open(os.path.join(od, "adapter_config.json")))
    assert acfg["r"] == 4 and "lm_head" not in acfg["target_modules"] and "q_proj" in acfg["target_modules"]

    # resuming restores the adapter and the connectors: the first loss of the resumed run sits near the end of the first run, not its start
    cfg2 = synth.write_train_fixture(str(tmp_path / "again"), SPECIAL_TOKENS, max_steps=1, resume_from_checkpoint=od,
                                     output_dir=str(tmp_path / "again" / "out"))
    torch.manual_seed(0)
    out2 = tr.run_train(cfg2)
    assert out2["log"][0]["loss"] < 0.5 * (losses[0] + min(losses[-3:])), (out2["log"][0]["loss"], losses)

    # the eval driver loads the checkpoint: adapter merged without peft, connectors from <output_dir>/connector
    from llamole_amd import eval as ev
    from llamole_amd import molecule_utils
    from llamole_amd.graph_decoder import GraphDiT
    from llamole_amd import graph_predictor
    from tests.cases import fake_template_runner
    monkeypatch.setattr(molecule_utils, "graph_to_smiles", lambda mols, dec: ["M" + "".join(chr(65 + int(a)) for a in at[:8]) for at, _ in mols])
    monkeypatch.setattr(GraphDiT, "check_valid", lambda self, s: True)
    monkeypatch.setattr(graph_predictor, "_default_template_runner", lambda: fake_template_runner)
    y = yaml.safe_load(open(cfg))
    ds = synth.write_molqa_dataset(os.path.join(str(tmp_path), "data"))
    y.update(adapter_name_or_path=od, graph_lm_connector_path=os.path.join(od, "connector"), do_train=False, dataset=ds, cutoff_len=32,
             max_new_tokens=8, per_device_eval_batch_size=2, output_dir=str(tmp_path / "evalout"))
    ycfg = str(tmp_path / "generate.yaml")
    yaml.safe_dump(y, open(ycfg, "w"))
    try:
        res = ev.run_eval(ycfg, overrides={"retro_iterations": 2, "retro_max_planning_time": 10})
    finally:
        from transformers.models.qwen2 import modeling_qwen2 as mq
        if hasattr(mq.apply_rotary_pos_emb, "_ll_orig"):
            mq.apply_rotary_pos_emb = mq.apply_rotary_pos_emb._ll_orig
    assert len(res["results"]) == 5 and all("llm_smiles" in r for r in res["results"])


def test_main_train_two_ranks(tmp_path):
    """Two processes on the one GPU of the box (gloo between them: one device cannot host two RCCL ranks): the data-parallel step --
    sharded micro batches, one bucketed gradient all-reduce per optimizer step -- through `python main.py train`'s own code."""
    from llamole_amd import synth
    from llamole_amd.modeling_llamole import SPECIAL_TOKENS
    cfg = synth.write_train_fixture(str(tmp_path), SPECIAL_TOKENS, num_train_epochs=3.0, per_device_train_batch_size=1)
    env = dict(os.environ, LLAMOLE_DIST_BACKEND="gloo", LLAMOLE_BENCH_SHARED_GPU="1", PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29613", os.path.join(ROOT, "tests", "train_rank_worker.py"), cfg]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("TRAIN_LOG ")][-1]
    log = json.loads(line[len("TRAIN_LOG "):])
    assert len(log["losses"]) == 3 and all(l == l for l in log["losses"])
    assert os.path.exists(os.path.join(str(tmp_path), "saves", "adapter", "adapter_model.safetensors"))


def test_main_train_trains_and_saves_the_new_token_embeddings(tmp_path, monkeypatch):
    """with a base vocabulary that lacks the Llamole special tokens the embedding matrices are RESIZED, and the reference
    then trains and saves them with the adapter (adapter.py:224-233: modules_to_save = {embed_tokens, lm_head} when resize_vocab and no
    additional_target) -- the rows of <design_body> / <retro_body> are the learned queries.  Here: the rows move during training, they are
    in the checkpoint, a second run with the same seed reproduces them bit for bit, and the eval-side loader ends up with exactly them."""
    from safetensors.torch import load_file
    from llamole_amd import eval as ev
    from llamole_amd import synth
    from llamole_amd import train as tr
    from llamole_amd.modeling_llamole import SPECIAL_TOKENS, GraphLLMForCausalMLM
    _fake_chem(monkeypatch)

    def run(tag, **kw):
        cfg = synth.write_train_fixture(str(tmp_path / tag), SPECIAL_TOKENS, exact_vocab=True, num_train_epochs=4.0, seed=7, **kw)
        out = tr.run_train(cfg)
        return cfg, out, load_file(os.path.join(out["output_dir"], "adapter_model.safetensors"))

    cfg0, out0, frozen = run("lr0", learning_rate=0.0)          # nothing moves: the checkpoint holds the seeded initial rows
    cfg1, out1, trained = run("lr", learning_rate=2.0e-3)
    _, _, again = run("lr_again", learning_rate=2.0e-3)
    assert sorted(m.split(".")[-1] for m in out1["modules_to_save"]) == ["embed_tokens", "lm_head"]
    ek = [k for k in trained if k.endswith("embed_tokens.weight")][0]
    hk = [k for k in trained if k.endswith("lm_head.weight")][0]
    model_args = ev.load_yaml_args(cfg1)[0]
    tok = ev.load_tokenizer(model_args)
    assert model_args.resize_vocab and trained[ek].shape[0] == len(tok) == trained[hk].shape[0]      # the resized matrices, not the base ones
    ids = [tok.convert_tokens_to_ids(t) for t in SPECIAL_TOKENS]
    body = [tok.convert_tokens_to_ids(t) for t in ("<design_body>", "<retro_body>")]
    kept = [tok.convert_tokens_to_ids(t) for t in ("<design_start>", "<retro_start>")]
    assert min(ids) >= trained[ek].shape[0] - len(SPECIAL_TOKENS)                                     # they are the new rows
    assert torch.equal(frozen[ek], load_file(os.path.join(out0["output_dir"], "adapter_model.safetensors"))[ek])
    d_in = (trained[ek].float() - frozen[ek].float()).abs().amax(dim=1)
    d_out = (trained[hk].float() - frozen[hk].float()).abs().amax(dim=1)
    assert all(float(d_in[i]) > 0 for i in body), d_in[body]          # the learned queries were trained
    assert all(float(d_out[i]) > 0 for i in kept), d_out[kept]        # <design_start> / <retro_start> stay in the labels: their output rows move
    for k in trained:                                                  # `seed` fixes the run: same seed, same checkpoint
        assert torch.equal(trained[k], again[k]), k
    # eval-side load: resize + merge ends with exactly the trained rows (a fresh resize alone would redraw them)
    import yaml
    y = yaml.safe_load(open(cfg1))
    y.update(adapter_name_or_path=out1["output_dir"], graph_lm_connector_path=os.path.join(out1["output_dir"], "connector"), do_train=False)
    ycfg = str(tmp_path / "gen.yaml")
    yaml.safe_dump(y, open(ycfg, "w"))
    margs, dargs, targs, fargs, _ = ev.load_yaml_args(ycfg)
    tok = ev.load_tokenizer(margs)
    m = GraphLLMForCausalMLM.from_pretrained(tok, margs, dargs, targs, fargs, load_adapter=True)
    w_in = m.language_model.get_input_embeddings().weight.detach().cpu()
    w_out = m.language_model.get_output_embeddings().weight.detach().cpu()
    assert torch.equal(w_in[ids], trained[ek][ids].to(w_in.dtype)) and torch.equal(w_out[ids], trained[hk][ids].to(w_out.dtype))


def test_main_train_with_a_gin_width_that_needs_padding(tmp_path, monkeypatch):
    """SFT with GIN encoder / predictor checkpoints of hidden_size 100 (zero-padded to 128 inside the engines): the HIP forward and the
    reverse sweep through the predictor run at the checkpoint's width, the loss is finite and moves."""
    from llamole_amd import synth
    from llamole_amd import train as tr
    from llamole_amd.modeling_llamole import SPECIAL_TOKENS
    _fake_chem(monkeypatch)
    cfg = synth.write_train_fixture(str(tmp_path), SPECIAL_TOKENS, gin_hidden=100, num_train_epochs=6.0, seed=3)
    out = tr.run_train(cfg)
    losses = [r["loss"] for r in out["log"]]
    assert len(losses) == 6 and all(l == l for l in losses) and all(r["retro_loss"] > 0 for r in out["log"])
    assert min(losses[-2:]) < losses[0], losses
