"""Deterministic stand-ins for the heavy parts around the orchestrator's host logic.

Used twice with identical inputs:
  * tests/golden/make_host_goldens.py drives the REFERENCE's own ``GraphLLMForCausalMLM`` methods
    (src/model/modeling_llamole.py:521-1287, loaded by file path in the build container) with these fakes and
    commits what they return / what they hand to the fakes -> tests/golden/host_traces.json;
  * tests/test_host_pins_cpu.py drives ``llamole_amd.modeling_llamole.GraphLLMForCausalMLM`` with the same fakes and
    compares against that file.
Nothing here is a restatement of reference logic: it is only the scripted world the logic runs in.
"""
import types

import torch

SPECIAL_TOKENS = ["<design_start>", "<design_end>", "<design_body>", "<molecule>", "<retro_start>", "<retro_end>",
                  "<retro_body>", "<rollback_start>", "<rollback_end>"]
TOKEN_IDS = {t: 70 + i for i, t in enumerate(SPECIAL_TOKENS)}
VOCAB, HIDDEN, GIN_H = 96, 16, 32


class Tok:
    """Character tokenizer (ids 3..52), special-token strings map to their ids; decode prints ids, specials as text."""
    eos_token_id = 0
    pad_token_id = 0

    def encode(self, text, add_special_tokens=False, return_tensors=None):
        ids, i = [], 0
        while i < len(text):
            hit = next((s for s in SPECIAL_TOKENS if text.startswith(s, i)), None)
            if hit is not None:
                ids.append(TOKEN_IDS[hit])
                i += len(hit)
            else:
                ids.append(3 + (ord(text[i]) % 50))
                i += 1
        return torch.tensor([ids]) if return_tensors == "pt" else ids

    def decode(self, ids, skip_special_tokens=False, **k):
        inv = {v: k_ for k_, v in TOKEN_IDS.items()}
        out = []
        for i in (ids.tolist() if torch.is_tensor(ids) else ids):
            i = int(i)
            if i in inv:
                if not skip_special_tokens:
                    out.append(inv[i])
            else:
                out.append(str(i))
        return " ".join(out)

    def apply_chat_template(self, messages, tokenize=False, add_generation_prompt=False):
        return "|".join(f"{m['role'][0]}:{m['content']}" for m in messages) + ("|a:" if add_generation_prompt else "")


class FakeLM(torch.nn.Module):
    """Embedding + running mean + linear head; ``generate`` replays a script of token rows (then a fixed pattern)."""

    def __init__(self, script=None):
        super().__init__()
        self.config = types.SimpleNamespace(hidden_size=HIDDEN, vocab_size=VOCAB)
        g = torch.Generator().manual_seed(0)
        self.emb = torch.nn.Embedding(VOCAB, HIDDEN)
        self.emb.weight.data = torch.randn(VOCAB, HIDDEN, generator=g)
        self.head = torch.nn.Linear(HIDDEN, VOCAB, bias=False)
        self.head.weight.data = torch.randn(VOCAB, HIDDEN, generator=g)
        self.model = types.SimpleNamespace(embed_tokens=self.emb)
        self.script = list(script or [])
        self.generate_calls, self.forward_calls = [], []

    def get_input_embeddings(self):
        return self.emb

    def forward(self, input_ids=None, attention_mask=None, output_hidden_states=False, return_dict=True, inputs_embeds=None,
                position_ids=None):
        self.forward_calls.append(None if input_ids is None else input_ids.clone())
        h = self.emb(input_ids) if inputs_embeds is None else inputs_embeds
        h = torch.cumsum(h, dim=1) / torch.arange(1, h.shape[1] + 1)[None, :, None]
        return types.SimpleNamespace(logits=self.head(h), hidden_states=(h, h))

    def generate(self, inputs=None, attention_mask=None, inputs_embeds=None, max_new_tokens=4, **kw):
        src = inputs if inputs is not None else inputs_embeds
        B = src.shape[0]
        self.generate_calls.append(dict(
            inputs=None if inputs is None else inputs.tolist(),
            embeds_sum=None if inputs_embeds is None else [round(float(v), 4) for v in inputs_embeds.detach().double().sum(dim=(1, 2))],
            embeds_len=None if inputs_embeds is None else int(inputs_embeds.shape[1]),
            max_new_tokens=max_new_tokens, kwargs={k: v for k, v in sorted(kw.items())}))
        if self.script:
            new = torch.tensor(self.script.pop(0), dtype=torch.long)
            assert new.shape[0] == B, (new.shape, B)
        else:
            n = min(int(max_new_tokens), 10)       # the budget itself is recorded above; keep the traces short
            new = torch.arange(B * n).reshape(B, n) % 7 + 20
        return new if inputs is None else torch.cat([inputs, new], dim=1)


class FakeDecoder:
    text_input_size = 768

    def __init__(self, smiles_script, invalid=()):
        self.smiles_script = [list(s) for s in smiles_script]
        self.invalid = set(invalid)
        self.generated = []

    def generate(self, props, cond, no_label):
        self.generated.append(dict(props=props.float().tolist(), cond=cond.float(), no_label=no_label))
        return list(self.smiles_script.pop(0))

    def check_valid(self, smiles):
        return smiles not in self.invalid

    def to(self, *a, **k):
        return self


class FakeEncoder(torch.nn.Module):
    hidden_size = GIN_H

    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(1)
        self.table = torch.nn.Parameter(torch.randn(120, GIN_H, generator=g), requires_grad=False)
        self.calls = []

    def forward(self, x, edge_index, edge_attr, batch):
        G = int(batch.max().item()) + 1
        self.calls.append(dict(x=x.tolist(), n_edges=int(edge_index.shape[1]), attr_sum=int(edge_attr.sum()), batch=batch.tolist()))
        out = torch.zeros(G, GIN_H).index_add_(0, batch, self.table[x])
        deg = torch.zeros(G).index_add_(0, batch[edge_index[1]], edge_attr.float()) if edge_index.shape[1] else torch.zeros(G)
        return out + 0.01 * deg[:, None]


class _Smiles:
    def __init__(self, items):
        self.items = list(items)

    def tolist(self):
        return list(self.items)


class FakePredictor:
    text_input_size = 768

    def __init__(self, table, available):
        self.table = table            # product smiles -> (reactants, scores, templates)
        self.available = {"smiles": _Smiles(available)}
        self.calls = []

    def sample_templates(self, product_graph, c, product_smiles, topk):
        self.calls.append(dict(product=product_smiles, x=product_graph.x.tolist(), cond=c.float(), topk=topk))
        r, s, t = self.table.get(product_smiles, ([], [], []))
        return list(r), list(s), list(t)

    def estimate_cost(self, smiles):
        return 0.25 * len(smiles)

    def to(self, *a, **k):
        return self


def fake_smiles_to_graph(data_cls):
    """SMILES string -> a ring graph over its characters (no rdkit in either image); '!' marks an unparsable string."""
    def fn(smiles):
        if "!" in smiles:
            return None
        x = torch.tensor([ord(c) % 100 for c in smiles[:6]], dtype=torch.long)
        n = x.numel()
        if n > 1:
            src = list(range(n)) + [(i + 1) % n for i in range(n)]
            dst = [(i + 1) % n for i in range(n)] + list(range(n))
            ei = torch.tensor([src, dst], dtype=torch.long)
            ea = torch.tensor([1 + (i % 4) for i in range(n)] * 2, dtype=torch.long)
        else:
            ei, ea = torch.empty((2, 0), dtype=torch.long), torch.empty((0,), dtype=torch.long)
        return data_cls(x=x, edge_index=ei, edge_attr=ea, num_nodes=n)
    return fn


def seeded_connectors(make):
    """The three Linear+SiLU connectors with seeded weights (make(d_in, d_out) -> nn.Sequential of the implementation)."""
    g = torch.Generator().manual_seed(7)
    out = {}
    for name, (i, o) in (("graph_to_lm_connector", (GIN_H, HIDDEN)), ("lm_to_graph_decoder", (HIDDEN, 768)),
                         ("lm_to_graph_predictor", (HIDDEN, 768))):
        seq = make(i, o)
        lin = seq[0]
        lin.weight.data = torch.randn(o, i, generator=g) * 0.2
        lin.bias.data = torch.randn(o, generator=g) * 0.1
        out[name] = seq
    return out


# ------------------------------------------------------------------------------------------ scenarios (inputs only)
S, B_, ME = TOKEN_IDS["<design_start>"], TOKEN_IDS["<design_body>"], TOKEN_IDS["<molecule>"]
RS, RB = TOKEN_IDS["<retro_start>"], TOKEN_IDS["<retro_body>"]
RBS, RBE = TOKEN_IDS["<rollback_start>"], TOKEN_IDS["<rollback_end>"]

BODY_CASES = [
    # (input rows, body id, n body, start id or None)
    ([[11, 12, S, 13, 14, 15, 16, 17, 18, 19, 21, 22], [31, 32, 33, 34, 35, 36, 37, 38, 39, 41, 42, 43]], B_, 8, S),
    ([[11, 12, S, 13, 14, S, 16, 17, 18, 19, 21, 22]], B_, 8, S),                  # two start tokens in a row
    ([[5, 6, 7]], B_, 8, S),                                                       # shorter than start + bodies
    ([[S, 6, 7, 8, 9, 10, 11, 12, 13, 14]], B_, 8, S),                             # start token first
    ([[1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, S]], B_, 8, S),    # start token last
    ([[31, 32, 33, 34, 35, 36], [1, 2, 3, 4, 5, 6]], RBS, 1, None),                # rollback form
    ([[4]], RBS, 1, None),
    ([[21, 22, 23, RS, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33]], RB, 8, RS),
]

# expansion table for the retrosynthesis scenarios: product -> (reactants, scores, templates)
RETRO_TABLE = {
    "CCO": (["CC.O", "C=C"], [0.7, 0.3], ["t_hydr", "t_red"]),
    "CC": (["C.C"], [1.0], ["t_couple"]),
    "C=C": ([], [], []),
    "NCC": (["N.CC", "NC!"], [0.6, 0.4], ["t_am", "t_bad"]),
}
AVAILABLE = ["O", "C", "N"]


# ------------------------------------------------------------------------------------------ scenario driver
def r4(t):
    return [round(float(v), 4) for v in torch.as_tensor(t).double().flatten()]


def cond_digest(c):
    """A condition tensor [B, 768] as a few numbers per row (sum, abs-sum, first 4 entries)."""
    c = c.detach().double()
    return [[round(float(r.sum()), 3), round(float(r.abs().sum()), 3)] + [round(float(v), 4) for v in r[:4]] for r in c]


def run_scenarios(build, Data, Batch, NO_LABEL_INDEX, IGNORE_INDEX):
    """Drive one GraphLLMForCausalMLM implementation (``build(lm_script, smiles_script, invalid)`` -> instance wired to the fakes
    above) through the scripted scenarios; returns plain data.  The SAME function produces the reference's pins and our values."""
    out = {"constants": {"NO_LABEL_INDEX": NO_LABEL_INDEX, "IGNORE_INDEX": IGNORE_INDEX}}

    # ---- add_special_body_tokens (a2)
    m = build()
    out["body_tokens"] = [m.add_special_body_tokens(torch.tensor(rows), body, n, start_token_id=start).tolist()
                          for rows, body, n, start in BODY_CASES]

    # ---- design_molecule (a2)
    prompt = [[5, 6, 7, 8, 9, 10], [11, 12, 13, 14, 15, 16]]
    props = [[0.5] + [float("nan")] * 9, [NO_LABEL_INDEX] * 10]
    analysis_script = [[21, 22, S, 23, 24, 25, 26, 27, 28, 29, 30, 31], [32, 33, 34, 35, 36, 37, 38, 39, 40, 41, 42, 43]]
    designs = {}
    for name, smiles, extra_script, rollback in [
            ("plain", [["CCO", "NCC"]], [], False),
            ("rollback_found", [["CCO", None]], [[[44, 45, RBE, 46]]], True),
            ("rollback_missing", [[None, None]], [[[44, 45, 46, 47], [48, RBE, 49, 50]]], True)]:
        m = build([analysis_script] + extra_script, smiles)
        analysis, sm = m.design_molecule(torch.tensor(prompt), torch.ones(2, 6, dtype=torch.long), torch.tensor(props), None, rollback,
                                         max_new_tokens=12, do_sample=False)
        lm, dec = m.language_model, m.graph_decoder
        designs[name] = dict(analysis=analysis.tolist(), smiles=sm, generate_calls=lm.generate_calls,
                             forward_ids=[f.tolist() for f in lm.forward_calls], decoder_props=str(dec.generated[0]["props"]),
                             decoder_no_label=dec.generated[0]["no_label"], cond=cond_digest(dec.generated[0]["cond"]))
    # molecule graphs spliced at <molecule>
    m = build([[[21, 22, 23, 24], [25, 26, S, 27]]], [["CCO", "CC"]])
    g = fake_smiles_to_graph(Data)
    graphs = Batch.from_data_list([g("CCN"), g("OCCO"), g("C")])
    ids = torch.tensor([[5, ME, 7, ME, 9, 10], [11, 12, 13, 14, ME, 16]])
    analysis, sm = m.design_molecule(ids, torch.ones_like(ids), torch.tensor(props), graphs, False, max_new_tokens=4)
    designs["with_graphs"] = dict(analysis=analysis.tolist(), smiles=sm, generate_calls=m.language_model.generate_calls,
                                  forward_ids=[f.tolist() for f in m.language_model.forward_calls], encoder_calls=m.graph_encoder.calls,
                                  cond=cond_digest(m.graph_decoder.generated[0]["cond"]))
    out["design"] = designs
    out["design_inputs"] = dict(prompt=prompt, analysis_script=analysis_script, mol_ids=ids.tolist())

    # ---- estimate_synthesis_complexity (a19)
    m = build()
    rx = lambda depth, template, mols: types.SimpleNamespace(depth=depth, template=template,                      # noqa: E731
                                                             children=[types.SimpleNamespace(mol=x) for x in mols])
    cases = [("CCO", None, 0, 1), ("c1ccccc1C(=O)O", (2, "[C:1]>>[C:1]O", ["CC", "N"]), 0.5, 1), ("CC(=O)N", (0, "T", []), 0, 2.0),
             ("O", (4, "A>>B.C", ["CCCCCCCC"]), 1.0, 0), ("N", None, None, None)]
    comp = []
    for smiles, r, mw, lw in cases:
        n0 = len(m.language_model.forward_calls)
        cost = m.estimate_synthesis_complexity(smiles, None, None if r is None else rx(*r), mw, lw)
        comp.append(dict(smiles=smiles, reaction=r, mol_w=mw, lang_w=lw, cost=cost,
                         prompt_ids=[f.tolist() for f in m.language_model.forward_calls[n0:]]))
    out["complexity"] = comp

    # ---- one_step_reaction (a18)
    steps = {}
    for name, product, with_ctx in [("no_context", "CCO", False), ("with_context", "NCC", True), ("invalid", "C!C", False)]:
        m = build([[[51, 52, RS, 53, 54, 55, 56, 57, 58, 59, 60, 61]]])
        ctx_ids = torch.tensor([5, ME, 7, 8]) if with_ctx else None
        ctx_graphs = Batch.from_data_list([g("OCC")]) if with_ctx else None
        res = m.one_step_reaction(product, ctx_ids, "Design text.", ctx_graphs, 7, max_new_tokens=9, do_sample=False)
        lm, pred = m.language_model, m.graph_predictor
        steps[name] = dict(result=res, generate_calls=lm.generate_calls, forward_ids=[f.tolist() for f in lm.forward_calls],
                           encoder_calls=m.graph_encoder.calls,
                           predictor_calls=[dict(product=c["product"], x=c["x"], topk=c["topk"], cond=cond_digest(c["cond"])) for c in pred.calls])
    out["one_step"] = steps

    # ---- generate (a1): design only, design + retrosynthesis (the reference's own molstar), invalid / unsolvable targets
    gens = {}
    m = build([analysis_script], [["CCO", "NCC"]])
    res = m.generate(torch.tensor(prompt), None, torch.tensor(props), do_molecular_design=True, do_retrosynthesis=False, max_new_tokens=12)
    res["design_analysis_tokens"] = res["design_analysis_tokens"].tolist()
    gens["design_only"] = res
    for name, smiles_in, invalid, rollback in [("retro_solved", ["CCO"], (), True), ("retro_invalid_target", ["XX"], ("XX",), True),
                                              ("retro_unsolved_rollback", ["C=C"], (), True), ("retro_unsolved_norollback", ["C=C"], (), False),
                                              ("retro_two", ["CCO", "CC"], (), True)]:
        m = build(None, [], invalid)
        ids2 = torch.tensor(prompt[:len(smiles_in)])
        res = m.generate(ids2, None, None, rollback=rollback, do_molecular_design=False, do_retrosynthesis=True,
                         input_smiles_list=list(smiles_in), iterations=6, max_planning_time=1e9, expansion_topk=5,
                         design_text_list=["Design text."], max_new_tokens=3)
        for plan in res["retro_plan_dict"].values():
            plan.pop("time", None)
        res["n_generate"] = len(m.language_model.generate_calls)
        res["n_forward"] = len(m.language_model.forward_calls)
        res["predictor_products"] = [c["product"] for c in m.graph_predictor.calls]
        gens[name] = res
    with torch.no_grad():
        try:
            build().generate(torch.tensor(prompt), None, None, do_molecular_design=False, do_retrosynthesis=False)
            gens["neither"] = "no error"
        except ValueError as e:
            gens["neither"] = "ValueError"
    out["generate"] = gens
    return out


def jsonable(o):
    if isinstance(o, dict):
        return {str(k): jsonable(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [jsonable(v) for v in o]
    if torch.is_tensor(o):
        return o.tolist()
    if isinstance(o, float) and o != o:
        return "nan"
    return o


# ---- SFT collator (reference src/data/collator.py:31-166): ragged features in the form the reference's preprocessing
# emits (processors/mmsupervised.py:262-312: input_ids, attention_mask, labels, molecule_ids, molecule_properties, retro_labels,
# retro_product_ids), for both padding sides and with pad_to_multiple_of
COLLATOR_MOLS = {3: "CCO", 5: "c1ccccc1", 8: "N", 11: "OCCN", 12: "CC(C)C"}


def collator_scenarios():
    f = lambda n, **k: dict(input_ids=list(range(10, 10 + n)), attention_mask=[1] * n, labels=[-100] * (n // 2) + list(range(50, 50 + n - n // 2)), **k)  # noqa: E731
    props = lambda i: [float(i), -200.0, 0.5 * i] + [float("nan")] * 0 + [1.0] * 7      # noqa: E731
    full = [f(7, molecule_ids=[3, 5], molecule_properties=props(1), retro_labels=[4, 9], retro_product_ids=[5, 11]),
            f(12, molecule_ids=[8], molecule_properties=props(2), retro_labels=[2], retro_product_ids=[12]),
            f(3, molecule_ids=[11, -100, 99, 12], molecule_properties=props(3), retro_labels=[7, 1, 0], retro_product_ids=[3, -100, 8]),
            f(9, molecule_ids=[], molecule_properties=props(4), retro_labels=[], retro_product_ids=[])]
    out = []
    for side in ("right", "left"):
        for mult in (None, 8):
            out.append({"name": f"full_{side}_{mult}", "padding_side": side, "pad_to_multiple_of": mult, "features": full})
    out.append({"name": "no_graph_keys", "padding_side": "right", "pad_to_multiple_of": None, "features": [f(4), f(6)]})
    out.append({"name": "first_only", "padding_side": "left", "pad_to_multiple_of": None,
                "features": [f(5, molecule_ids=[12, 3], retro_labels=[6], retro_product_ids=[5]), f(2)]})
    return out


def collator_record(batch):
    """The fields both collators produce, as plain data."""
    rec = {}
    for k in ("input_ids", "attention_mask", "labels", "retro_labels", "molecule_properties"):
        v = batch.get(k) if hasattr(batch, "get") else None
        rec[k] = None if v is None else (v.tolist() if torch.is_tensor(v) else v)
    for k in ("molecule_graphs", "design_graphs", "retro_product_graphs"):
        g = batch.get(k) if hasattr(batch, "get") else None
        rec[k] = None if g is None else {"x": g.x.tolist(), "edge_index": g.edge_index.tolist(), "edge_attr": g.edge_attr.tolist(),
                                         "batch": g.batch.tolist(), "num_graphs": int(g.num_graphs)}
    return rec


# ---- SFT forward (reference modeling_llamole.py:299-437) on a REAL tiny HF causal LM with differentiable stand-ins for
# the graph modules: what enters the loss, which hidden states feed the retro queries, what the connectors' gradients are
class SFTPred(torch.nn.Module):
    text_input_size, available = 768, None

    def __init__(self):
        super().__init__()
        self.register_buffer("W", torch.randn(7, 768, generator=torch.Generator().manual_seed(3)) * 0.05)

    def forward(self, x, ei, ea, batch, c):
        pooled = torch.stack([x[batch == g].float().sum() for g in range(int(batch.max()) + 1)])
        return c.float() @ self.W.t() + 0.01 * pooled[:, None]


class SFTEnc(torch.nn.Module):
    hidden_size = 32

    def forward(self, x, ei, ea, b):
        return torch.stack([x[b == g].float().mean().repeat(32) for g in range(int(b.max()) + 1)]) * 0.01


class SFTDec(torch.nn.Module):
    """GraphDiT.forward stand-in: a loss that depends on its condition (the reference computes it and then drops it, :359-381, :421-425)."""
    text_input_size = 768

    def forward(self, x, ei, ea, batch, props, cond, no_label):
        return cond.float().pow(2).mean() + 3.0


def sft_forward_case(Data, Batch, token_ids, IGNORE_INDEX, NO_LABEL_INDEX):
    g = torch.Generator().manual_seed(0)
    B, L = 2, 40
    ids = torch.randint(5, 1000, (B, L), generator=g)
    ids[0, 3] = token_ids["<molecule>"]
    ids[1, 5] = token_ids["<molecule>"]
    for b, start in ((0, 10), (0, 25), (1, 12)):
        ids[b, start] = token_ids["<retro_start>"]
        ids[b, start + 1:start + 9] = token_ids["<retro_body>"]
    ids[1, 24] = token_ids["<design_start>"]
    ids[1, 25:33] = token_ids["<design_body>"]
    labels = ids.clone()
    labels[:, :4] = IGNORE_INDEX
    retro_labels = torch.tensor([[4, NO_LABEL_INDEX], [2, IGNORE_INDEX]])
    mk = lambda n, k: Data(x=torch.arange(n) % 9 + k, edge_index=torch.empty((2, 0), dtype=torch.long), edge_attr=torch.empty((0,), dtype=torch.long), num_nodes=n)  # noqa: E731
    return dict(input_ids=ids, attention_mask=torch.ones_like(ids), labels=labels, molecule_graphs=Batch.from_data_list([mk(4, 1), mk(6, 2)]),
                molecule_properties=torch.tensor([[1.0] * 10, [2.0] * 10]), design_graphs=Batch.from_data_list([mk(5, 2)]),
                retro_labels=retro_labels, retro_product_graphs=Batch.from_data_list([mk(5, 3), mk(3, 1), mk(7, 2)]))


def sft_connectors(hidden: int):
    g = torch.Generator().manual_seed(11)
    out = {}
    for name, (i, o) in (("graph_to_lm_connector", (32, hidden)), ("lm_to_graph_decoder", (hidden, 768)), ("lm_to_graph_predictor", (hidden, 768))):
        lin = torch.nn.Linear(i, o)
        with torch.no_grad():
            lin.weight.copy_(torch.randn(o, i, generator=g) * 0.05)
            lin.bias.copy_(torch.randn(o, generator=g) * 0.05)
        out[name] = torch.nn.Sequential(lin, torch.nn.SiLU())
    return out


def sft_forward_record(model, batch):
    """loss, logits digest and gradient norms of one forward + backward of `model` (reference or llamole_amd instance)."""
    out = model(**batch)
    loss = out.loss if hasattr(out, "loss") else out["loss"]
    logits = out.logits if hasattr(out, "logits") else out["logits"]
    model.zero_grad(set_to_none=True)
    loss.backward()
    grads = {n: float(p.grad.double().norm()) for n, p in model.named_parameters() if p.grad is not None and
             (n.startswith(("graph_to_lm_connector", "lm_to_graph_decoder", "lm_to_graph_predictor")) or n.endswith(("layers.0.self_attn.q_proj.weight", "lm_head.weight")))}
    return {"loss": float(loss.detach()), "logits_sum": float(logits.detach().double().sum()), "logits_abs": float(logits.detach().double().abs().sum()),
            "grad_norms": {_llm_suffix(k): v for k, v in sorted(grads.items())}}


def _llm_suffix(name: str) -> str:
    """Parameter name without the wrapper prefixes that differ between the two instances (peft shape | plain HF model)."""
    for key in ("layers.0.", "lm_head."):
        if key in name:
            return name[name.index(key):]
    return name


# ---- eval dataset rows (reference src/eval/dataset.py:26-78)
MOLQA_RECORDS = [
    {"instruction": "Design a polymer with high CO2 permeability.", "input": "It should contain an ether linkage.",
     "property": {"CO2": 120.5, "O2": 33.0, "SA": 2.5}},
    {"instruction": "Propose a drug-like molecule.", "input": "", "property": {"BBBP": 1.0, "HIV": 0.0, "BACE": 1.0, "SC": 3.25}},
    {"instruction": "Make it " + "very " * 60 + "long.", "input": "Truncation applies to this row.", "property": {}},
]
