"""Interleaved-generation orchestrator: counterpart of reference
``src/model/modeling_llamole.py:GraphLLMForCausalMLM`` for the MI355X graph engines.

Host code stays Python on PyTorch-ROCm and the HuggingFace LLM is untouched (north star); this module owns
the control flow between the LLM and the three graph engines:

  generate()                      reference :1115-1287   dispatch design / retrosynthesis, assemble the info dict
  design_molecule()               :584-663   LLM decode to the trigger -> query tokens -> GraphDiT (HIP)
  add_special_body_tokens()       :521-582   [kept analysis][<x_start>][8 x <x_body>] left-padded with eos
  one_step_reaction()             :784-889   GIN-encode product (HIP) -> LLM decode -> GIN predictor (HIP) -> templates
  estimate_synthesis_complexity() :891-993   LLM-logit cost (+ optional CostMLP on HIP)
  retrosynthesize()               :995-1093  A* over one_step_reaction / estimate_synthesis_complexity
  connectors                      :205-222   Linear+SiLU x3, stored as connector/*.pt

Reference quirks kept on purpose (drivers depend on the shapes of the results): ``design_text_list[0]`` is
used for every batch element (:1174-1175); ``retro_plan_dict`` is keyed by SMILES (:1178); the SFT loss adds the
retro loss under both the design and the retro weight (:421-425).
"""
from __future__ import annotations

import json
import os
import time
from typing import Any, Dict, List, Optional, Union

import torch
import torch.nn as nn

from .graph_data import GraphBatch, GraphData
from .planner import molstar

IGNORE_INDEX = -100      # reference src/extras/constants.py:24
NO_LABEL_INDEX = -200    # :25
BOND_INDEX_BY_NAME = {"SINGLE": 1, "DOUBLE": 2, "TRIPLE": 3, "AROMATIC": 4}   # :51 (keyed by rdkit BondType there)

SPECIAL_TOKENS = ["<design_start>", "<design_end>", "<design_body>", "<molecule>", "<retro_start>", "<retro_end>",
                  "<retro_body>", "<rollback_start>", "<rollback_end>"]


class GraphLMOutput(dict):
    """Result of the SFT forward (reference GraphLMOutput, :40-48): dict with attribute access; ``loss`` carries the graph."""
    __getattr__ = dict.get


def make_connector(d_in: int, d_out: int) -> nn.Sequential:
    return nn.Sequential(nn.Linear(d_in, d_out), nn.SiLU())


class GraphLLMForCausalMLM(nn.Module):
    def __init__(self, model_args, finetuning_args, data_args, language_model, graph_decoder, graph_predictor,
                 graph_encoder, token_id_dict, tokenizer):
        super().__init__()
        self.language_model = language_model
        self.graph_decoder = graph_decoder
        self.graph_predictor = graph_predictor
        self.graph_encoder = graph_encoder
        self.token_id_dict = token_id_dict
        self.num_body_tokens = getattr(data_args, "learned_query_size", 8)
        self.model_args, self.finetuning_args, self.data_args = model_args, finetuning_args, data_args
        self.tokenizer = tokenizer
        self.config = getattr(language_model, "config", None)
        hidden = self.config.hidden_size
        self.graph_to_lm_connector = make_connector(graph_encoder.hidden_size, hidden)
        self.lm_to_graph_decoder = make_connector(hidden, graph_decoder.text_input_size)
        self.lm_to_graph_predictor = make_connector(hidden, graph_predictor.text_input_size)
        self.timings: Dict[str, float] = {}
        self.decoder = None      # optional llm_decode.GraphedDecoder (HIP-graph decode step); None = HF generate
        self.reuse_query_kv = False
        self.retro_max_new_tokens = 512   # analysis budget of one expansion (the reference hard-codes it, :846-848)
        self.batch_retro = False   # lock-step A* searches of one batch with batched expansions (retrosynthesize_many)
        self.batch_values = True   # A* value estimates of one expansion in one LLM forward (estimate_synthesis_complexity_batch)

    def enable_graphed_decode(self, use_graph: bool = True, sync_every: int = 16, fused_cache: bool = False,
                              reuse_query_kv: bool = True):
        """Route every LLM decode of the path through one captured hipGraph of the stock HF forward; ``reuse_query_kv``
        runs the query-token re-forward on top of the decode's KV cache whenever that is the same computation."""
        from .llm_decode import GraphedDecoder
        self.decoder = GraphedDecoder(self.language_model, use_graph=use_graph, sync_every=sync_every, fused_cache=fused_cache)
        self.reuse_query_kv = reuse_query_kv
        return self

    def enable_mi355x_decode(self, use_graph: bool = True, **kw) -> dict:
        """The whole LLM-side stack in one call (what bench.py measures): HIP kernels under the HF modules
        (llm_accel.accelerate_llm) + hipGraph decode with the fused KV append and in-graph sampler.  Returns the report of
        what took effect; on a CPU model it only installs the (eager) static-cache decoder."""
        from .llm_accel import accelerate_llm
        info = accelerate_llm(self.language_model)
        self.enable_graphed_decode(use_graph=use_graph, fused_cache=bool(info.get("decode_attention")), **kw)
        return info

    def _llm_generate(self, inputs=None, attention_mask=None, inputs_embeds=None, **kwargs):
        if self.decoder is not None:
            return self.decoder.generate(input_ids=inputs, attention_mask=attention_mask, inputs_embeds=inputs_embeds, **kwargs)
        if inputs_embeds is not None:
            return self.language_model.generate(attention_mask=attention_mask, inputs_embeds=inputs_embeds, **kwargs)
        return self.language_model.generate(inputs=inputs, attention_mask=attention_mask, **kwargs)

    # ------------------------------------------------------------------ construction
    @classmethod
    def from_pretrained(cls, tokenizer, model_args, data_args, training_args, finetuning_args, load_adapter=False,
                        add_valuehead=False, language_model=None):
        """Reference :102-286.  The LLM itself is stock HuggingFace (+ peft adapter) on PyTorch-ROCm; pass
        ``language_model`` to reuse an already-built one."""
        from .loader import load_graph_decoder, load_graph_encoder, load_graph_predictor
        compute_dtype = getattr(model_args, "compute_dtype", torch.bfloat16)
        if load_adapter:
            apath = getattr(model_args, "adapter_name_or_path", None)
            if apath is None:
                raise ValueError("Please specify the adapter_name_or_path when load_adapter is True.")
            if len(apath) != 1:
                raise ValueError("Only one adapter is supported at a time.")
        if language_model is None:
            from transformers import AutoModelForCausalLM
            language_model = AutoModelForCausalLM.from_pretrained(model_args.model_name_or_path, torch_dtype=compute_dtype)
            if len(tokenizer) > language_model.get_input_embeddings().weight.shape[0]:
                language_model.resize_token_embeddings(len(tokenizer))
            if load_adapter:
                try:
                    from peft import PeftModel
                    language_model = PeftModel.from_pretrained(language_model, model_args.adapter_name_or_path[0]).merge_and_unload()
                except ImportError:       # no peft in this image: merge the LoRA adapter from its on-disk layout directly
                    from .sft import merge_lora_adapter
                    merge_lora_adapter(language_model, model_args.adapter_name_or_path[0])
            language_model.to("cuda").eval()
        device = next(language_model.parameters()).device
        graph_decoder = load_graph_decoder(model_args, path=model_args.graph_decoder_path, device=device)
        graph_predictor = load_graph_predictor(model_args, path=model_args.graph_predictor_path, device=device)
        graph_encoder = load_graph_encoder(model_args, path=model_args.graph_encoder_path, device=device)
        token_id_dict = {}
        for elem in getattr(model_args, "new_special_tokens", None) or SPECIAL_TOKENS:
            if isinstance(elem, str) and len(elem) != 0:
                token_id_dict[elem] = tokenizer.encode(elem, add_special_tokens=False)[0]
        model = cls(model_args, finetuning_args, data_args, language_model, graph_decoder, graph_predictor,
                    graph_encoder, token_id_dict, tokenizer)
        for conn in (model.graph_to_lm_connector, model.lm_to_graph_decoder, model.lm_to_graph_predictor):
            for p in conn.parameters():
                if p.dtype == torch.float32 and compute_dtype != torch.float32:
                    p.data = p.data.to(compute_dtype)
        if load_adapter:
            cpath = getattr(model_args, "graph_lm_connector_path", None)
            if not cpath:
                raise ValueError("Connector should be downloaded with the adapter: set graph_lm_connector_path")
            for name in ("graph_to_lm_connector", "lm_to_graph_decoder", "lm_to_graph_predictor"):
                getattr(model, name).load_state_dict(torch.load(os.path.join(cpath, name + ".pt"), map_location=device, weights_only=True))
        for name in ("graph_to_lm_connector", "lm_to_graph_decoder", "lm_to_graph_predictor"):
            getattr(model, name).to(device)
        return model

    def save_pretrained(self, save_directory, save_graph_modules: bool = False, modules_to_save=(), **kwargs):
        """Reference :439-519: the language model (its ADAPTER when it carries one -- peft layout, ``save_peft_format``), optionally the
        three graph modules, the three connectors under ``connector/`` and ``graphllm_config.json``."""
        if os.path.isfile(save_directory):
            raise ValueError(f"Provided path ({save_directory}) should be a directory, not a file")
        os.makedirs(save_directory, exist_ok=True)
        from .sft import LoRALinear, save_lora_adapter
        if any(isinstance(m, LoRALinear) for m in self.language_model.modules()):
            save_lora_adapter(self.language_model, save_directory, modules_to_save=modules_to_save,
                              base_model_name=getattr(self.model_args, "model_name_or_path", None))
        elif hasattr(self.language_model, "save_pretrained") and kwargs.get("save_language_model", True):
            self.language_model.save_pretrained(save_directory)
        if save_graph_modules:
            for name in ("graph_decoder", "graph_predictor", "graph_encoder"):
                mod = getattr(self, name)
                if hasattr(mod, "save_pretrained"):
                    mod.save_pretrained(os.path.join(save_directory, name))
        os.makedirs(os.path.join(save_directory, "connector"), exist_ok=True)
        for name in ("graph_to_lm_connector", "lm_to_graph_decoder", "lm_to_graph_predictor"):
            torch.save(getattr(self, name).state_dict(), os.path.join(save_directory, "connector", name + ".pt"))

        def plain(ns):
            out = {}
            for k, v in (vars(ns) if hasattr(ns, "__dict__") else {}).items():
                if isinstance(v, (str, int, float, bool, type(None))):
                    out[k] = v
                elif isinstance(v, (list, tuple)) and all(isinstance(x, (str, int, float, bool)) for x in v):
                    out[k] = list(v)
                elif isinstance(v, torch.dtype):
                    out[k] = str(v)
            return out
        w = self.finetuning_args
        with open(os.path.join(save_directory, "graphllm_config.json"), "w") as f:
            json.dump({"model_args": plain(self.model_args), "finetuning_args": plain(self.finetuning_args), "data_args": plain(self.data_args),
                       "token_id_dict": self.token_id_dict, "num_body_tokens": self.num_body_tokens,
                       "loss_weight_lm": getattr(w, "loss_weight_lm", 1), "loss_weight_design": getattr(w, "loss_weight_design", 1),
                       "loss_weight_retro": getattr(w, "loss_weight_retro", 1)}, f, indent=2)

    @property
    def device(self):
        return next(self.language_model.parameters()).device

    def forward(self, input_ids=None, attention_mask=None, labels=None, molecule_graphs=None, molecule_properties=None,
                design_graphs=None, retro_labels=None, retro_product_graphs=None, past_key_values=None, use_cache=None,
                output_attentions=None, output_hidden_states=True, return_dict=None):
        """SFT forward (reference :299-437): next-token loss of the HF LLM over the prompt with GIN-encoded
        molecules spliced in, plus the retrosynthesis cross-entropy of the frozen GIN predictor conditioned on the mean
        hidden state of the ``<retro_body>`` query tokens.  Gradients reach the LLM (adapter) and the connectors; through
        the predictor they are the HIP reverse sweep (GraphPredictor.forward with ``c.requires_grad``).

        ``total = w_lm * lm_loss + (w_design + w_retro) * retro_loss`` -- the reference multiplies ``loss_weight_design``
        by the retro loss as well (:421-425) and never adds its design loss, so the GraphDiT training forward it runs
        (:359-381) changes neither the loss nor any gradient; it is skipped here unless ``self.compute_design_loss``."""
        emb_layer = self.language_model.get_input_embeddings()
        inputs_embeds = emb_layer(input_ids)
        mol_pos = (input_ids == self.token_id_dict["<molecule>"]).nonzero()
        if molecule_graphs is not None and mol_pos.shape[0] > 0:
            with torch.no_grad():          # the encoder is frozen and its inputs are integers
                mol = self.graph_encoder(molecule_graphs.x, molecule_graphs.edge_index, molecule_graphs.edge_attr,
                                         molecule_graphs.batch)
            mol = self.graph_to_lm_connector(mol.to(next(self.graph_to_lm_connector.parameters()).dtype))
            assert mol_pos.shape[0] == mol.shape[0], \
                f"Number of molecule tokens ({mol_pos.shape[0]}) does not match number of molecule embeddings ({mol.shape[0]})"
            inputs_embeds = inputs_embeds.clone()
            inputs_embeds[mol_pos[:, 0], mol_pos[:, 1]] = mol.to(inputs_embeds.dtype)
        lm_out = self.language_model(input_ids=None, attention_mask=attention_mask, past_key_values=past_key_values,
                                     inputs_embeds=inputs_embeds, use_cache=use_cache, output_attentions=output_attentions,
                                     output_hidden_states=True, return_dict=True, labels=labels)
        lm_loss = lm_out.loss if lm_out.loss is not None else 0
        hidden = lm_out.hidden_states[-1]
        body = torch.arange(self.num_body_tokens, device=input_ids.device)

        design_loss = 0
        if design_graphs is not None and getattr(self, "compute_design_loss", False):
            pos = (input_ids == self.token_id_dict["<design_start>"]).nonzero()
            if pos.numel() > 0:
                dh = hidden[pos[:, 0].unsqueeze(1), (pos[:, 1] + 1).unsqueeze(1) + body].mean(dim=1)
                dh = self.lm_to_graph_decoder(dh.to(next(self.lm_to_graph_decoder.parameters()).dtype))
                design_loss = self.graph_decoder(design_graphs.x, design_graphs.edge_index, design_graphs.edge_attr,
                                                 design_graphs.batch, molecule_properties, dh, NO_LABEL_INDEX)

        retro_loss = 0
        if retro_labels is not None:
            pos = (input_ids == self.token_id_dict["<retro_start>"]).nonzero()
            flat = retro_labels[retro_labels != IGNORE_INDEX]
            valid = flat != NO_LABEL_INDEX
            pos, flat = pos[valid], flat[valid]
            if len(flat) > 0:
                rh = hidden[pos[:, 0].unsqueeze(1), (pos[:, 1] + 1).unsqueeze(1) + body].mean(dim=1)
                graphs = retro_product_graphs.to_data_list() if hasattr(retro_product_graphs, "to_data_list") else list(retro_product_graphs)
                keep = valid.nonzero().view(-1).tolist()
                gb = GraphBatch.from_data_list([graphs[i] for i in keep])
                rh = self.lm_to_graph_predictor(rh.to(next(self.lm_to_graph_predictor.parameters()).dtype))
                pred = self.graph_predictor(gb.x, gb.edge_index, gb.edge_attr, gb.batch, rh)
                retro_loss = torch.nn.functional.cross_entropy(pred.float(), flat.to(pred.device))

        w = self.finetuning_args
        total = (getattr(w, "loss_weight_lm", 1) * lm_loss + getattr(w, "loss_weight_design", 1) * retro_loss
                 + getattr(w, "loss_weight_retro", 1) * retro_loss)
        out = {"loss": total, "logits": lm_out.logits, "past_key_values": getattr(lm_out, "past_key_values", None),
               "hidden_states": lm_out.hidden_states, "attentions": getattr(lm_out, "attentions", None),
               "additional_log_info": {k: (v.detach() if torch.is_tensor(v) else v) for k, v in
                                       (("lm_loss", lm_loss), ("retro_loss", retro_loss), ("design_loss", design_loss))}}
        if return_dict is False:
            return (total, lm_out.logits, out["past_key_values"], lm_out.hidden_states)
        return GraphLMOutput(out)

    # ------------------------------------------------------------------ helpers
    def add_special_body_tokens(self, input_ids, body_token_id, num_body_tokens, start_token_id=None):
        """Append ``[start?] + num_body_tokens x body`` after the (first) start token of every row -- or at the
        end if the row has none -- keeping the rightmost context that fits, left-padded with eos (:521-582)."""
        bsz, seq_len = input_ids.shape
        start_len = 1 if start_token_id is not None else 0
        if seq_len < num_body_tokens + start_len:
            seq_len = seq_len + num_body_tokens + start_len
        dev = input_ids.device
        cut = [seq_len - start_len - num_body_tokens] * bsz
        keep_budget = seq_len - num_body_tokens
        if start_token_id is not None:
            rows, cols = (input_ids == start_token_id).nonzero(as_tuple=True)
            for r, c in zip(rows.tolist(), cols.tolist()):
                cut[r] = c                      # last occurrence wins, as in the reference loop
            keep_budget = seq_len - num_body_tokens - 1
        out = torch.full((bsz, seq_len), self.tokenizer.eos_token_id, device=dev, dtype=input_ids.dtype)
        body = torch.full((num_body_tokens,), body_token_id, device=dev, dtype=input_ids.dtype)
        for i in range(bsz):
            lo = max(0, cut[i] - keep_budget)
            parts = [input_ids[i, lo:cut[i]]]
            if start_token_id is not None:
                parts.append(torch.tensor([start_token_id], device=dev, dtype=input_ids.dtype))
            parts.append(body)
            tail = torch.cat(parts)
            out[i, seq_len - tail.numel():] = tail
        return out

    def _query_hidden(self, ids: torch.Tensor) -> torch.Tensor:
        """LLM forward over ids (all-ones mask, as the reference) -> mean of the last num_body_tokens hidden states."""
        kw = dict(input_ids=ids, attention_mask=torch.ones_like(ids), output_hidden_states=True, return_dict=True)
        try:      # only hidden states are used: skip the [B, L, vocab] lm_head product where the model allows it
            out = self.language_model(logits_to_keep=1, **kw)
        except TypeError:
            out = self.language_model(**kw)
        return out.hidden_states[-1][:, -self.num_body_tokens:].mean(dim=1)

    def _query_hidden_from_cache(self, input_ids, attention_mask, analysis, design_ids):
        """KV-cache reuse for the query-token re-forward.  Valid only when the token sequence of the
        re-forward is the decoded sequence with its last tokens replaced by ``[<design_start>] + bodies``: full-length
        analysis without a ``<design_start>`` trigger (add_special_body_tokens then keeps ``analysis[:-9]`` in place) and an
        unpadded prompt (the re-forward uses an all-ones mask).  Returns None when the full re-forward is required."""
        dec = self.decoder
        info = getattr(dec, "_last", None) if dec is not None else None
        tail = self.num_body_tokens + 1
        L = analysis.shape[1]
        if (info is None or not info["from_ids"] or info.get("n_new") != info["max_new"] or L != info["max_new"] or L <= tail
                or design_ids.shape[1] != input_ids.shape[1] + L):
            return None
        start_id = self.token_id_dict["<design_start>"]
        if bool((analysis == start_id).any()) or not bool(attention_mask.bool().all()):
            return None
        hs = dec.continue_hidden(design_ids[:, -tail:], input_ids.shape[1] + L - tail)
        return hs[:, -self.num_body_tokens:].mean(dim=1)

    def _splice_molecules(self, ids: torch.Tensor, graphs) -> torch.Tensor:
        """embed_tokens(ids) with every <molecule> position replaced by connector(GIN encoder(graph)) (:607-622)."""
        emb_layer = self.language_model.get_input_embeddings()
        inputs_embeds = emb_layer(ids)
        pos = (ids == self.token_id_dict["<molecule>"]).nonzero()
        mol = self.graph_encoder(graphs.x, graphs.edge_index, graphs.edge_attr, graphs.batch)
        mol = self.graph_to_lm_connector(mol.to(next(self.graph_to_lm_connector.parameters()).dtype))
        assert pos.shape[0] == mol.shape[0], \
            f"Number of molecule tokens ({pos.shape[0]}) does not match number of molecule embeddings ({mol.shape[0]})"
        inputs_embeds[pos[:, 0], pos[:, 1]] = mol.to(inputs_embeds.dtype)
        return inputs_embeds

    # ------------------------------------------------------------------ design
    @torch.no_grad()
    def design_hidden(self, input_ids, attention_mask, molecule_graphs=None, **kwargs):
        """Steps 1-3 of design_molecule: analysis tokens and the [B,768] text condition for GraphDiT."""
        t0 = time.perf_counter()
        if molecule_graphs is None:
            analysis = self._llm_generate(inputs=input_ids, attention_mask=attention_mask, **kwargs)
            analysis = analysis[:, input_ids.shape[1]:]
        else:
            embeds = self._splice_molecules(input_ids, molecule_graphs)
            analysis = self._llm_generate(attention_mask=attention_mask, inputs_embeds=embeds, **kwargs)
        t1 = time.perf_counter()
        from ._trace import mark
        design_ids = self.add_special_body_tokens(analysis, self.token_id_dict["<design_body>"], self.num_body_tokens,
                                                  start_token_id=self.token_id_dict["<design_start>"])
        design_ids = torch.cat([input_ids, design_ids], dim=1)
        mark("design: body tokens added")
        hidden = None
        if molecule_graphs is None and self.reuse_query_kv:
            hidden = self._query_hidden_from_cache(input_ids, attention_mask, analysis, design_ids)
        if hidden is None:
            hidden = self._query_hidden(design_ids)
        mark("design: query forward enqueued")
        cond = self.lm_to_graph_decoder(hidden.to(next(self.lm_to_graph_decoder.parameters()).dtype))
        mark("design: connector enqueued")
        self.timings.update(llm_decode_s=t1 - t0, llm_query_s=time.perf_counter() - t1)
        return analysis, design_ids, cond

    @torch.no_grad()
    def design_molecule(self, input_ids, attention_mask, molecule_properties=None, molecule_graphs=None,
                        rollback=False, **kwargs):
        analysis, design_ids, cond = self.design_hidden(input_ids, attention_mask, molecule_graphs, **kwargs)
        t0 = time.perf_counter()
        smiles_list = self.graph_decoder.generate(molecule_properties.to(cond.dtype), cond, NO_LABEL_INDEX)
        self.timings["graphdit_s"] = time.perf_counter() - t0
        if rollback and None in smiles_list:
            smiles_list = self.design_rollback(design_ids, smiles_list, **kwargs)
        return analysis, smiles_list

    def design_rollback(self, analysis_tokens, smiles_list, **kwargs):
        """GraphDiT produced an invalid molecule: let the LLM write the SMILES itself (:665-718)."""
        none_idx = [i for i, s in enumerate(smiles_list) if s is None]
        if not none_idx:
            return smiles_list
        rb_ids = self.add_special_body_tokens(analysis_tokens[torch.tensor(none_idx)], self.token_id_dict.get("<rollback_start>"), 1)
        if "max_new_tokens" in kwargs:
            kwargs["max_new_tokens"] *= 2
        new_tokens = self._llm_generate(inputs=rb_ids, attention_mask=torch.ones_like(rb_ids), **kwargs)
        end_text = self.tokenizer.decode([self.token_id_dict.get("<rollback_end>")])
        for i, seq in zip(none_idx, new_tokens[:, rb_ids.shape[1]:]):
            text = self.tokenizer.decode(seq, skip_special_tokens=False)
            cut = text.find(end_text)
            smiles_list[i] = text[:cut].strip() if cut != -1 else None
        return smiles_list

    # ------------------------------------------------------------------ retrosynthesis
    def smiles_to_graph(self, smiles: str) -> Optional[GraphData]:
        """SMILES -> integer graph (x = atomic number - 2, '*' -> 117; symmetric edges; bond classes 1..4) (:720-760)."""
        try:
            from rdkit import Chem
        except ImportError as e:  # pragma: no cover
            raise ImportError("smiles_to_graph needs `rdkit` (reference requirements.txt:22)") from e
        mol = Chem.MolFromSmiles(smiles)
        if mol is None:
            return None
        x = torch.tensor([117 if a.GetSymbol() == "*" else a.GetAtomicNum() - 2 for a in mol.GetAtoms()
                          if a.GetAtomicNum() != 1], dtype=torch.long)
        src, dst, typ = [], [], []
        for b in mol.GetBonds():
            i, j = b.GetBeginAtomIdx(), b.GetEndAtomIdx()
            if mol.GetAtomWithIdx(i).GetAtomicNum() != 1 and mol.GetAtomWithIdx(j).GetAtomicNum() != 1:
                src += [i, j]
                dst += [j, i]
                typ += [BOND_INDEX_BY_NAME.get(str(b.GetBondType()), 1)] * 2
        if src:
            return GraphData(x, torch.tensor([src, dst], dtype=torch.long), torch.tensor(typ, dtype=torch.long))
        return GraphData(x, torch.empty((2, 0), dtype=torch.long), torch.empty((0,), dtype=torch.long))

    def retrosynthesize_rollback(self, input_ids, design_text, smiles, **kwargs):
        """Planning failed / invalid target: free-text synthesis by the LLM (:762-782)."""
        ids = self.tokenizer.encode(f"{design_text} To synthesize {smiles}, follow these procedures: ",
                                    add_special_tokens=False, return_tensors="pt").to(self.device)
        if "max_new_tokens" in kwargs:
            kwargs["max_new_tokens"] = 256
        out = self._llm_generate(inputs=ids, **kwargs)[:, ids.shape[1]:]
        return self.tokenizer.encode(f"To synthesize {smiles}, follow these procedures: ") + out.cpu().squeeze().tolist()

    def one_step_reaction(self, product_smiles, input_ids, design_text, molecule_graphs, topk, **kwargs):
        prompt = self.tokenizer.encode(f"{design_text} To synthesize <molecule>, follow these procedures: ",
                                       add_special_tokens=False, return_tensors="pt").to(self.device)
        with_context = input_ids is not None and molecule_graphs is not None
        if with_context:
            prompt = torch.cat([input_ids.view(1, -1), prompt], dim=-1)
        product = self.smiles_to_graph(product_smiles)
        if product is None:
            return {"reactants": [], "scores": [], "templates": [],
                    "analysis": self.tokenizer.encode("Invalid product SMILES", add_special_tokens=False)}
        product.to(self.device)
        graphs = GraphBatch.from_data_list((molecule_graphs.to_data_list() if with_context else []) + [product])
        embeds = self._splice_molecules(prompt, graphs)
        if "max_new_tokens" in kwargs:
            kwargs["max_new_tokens"] = self.retro_max_new_tokens
        analysis = self._llm_generate(attention_mask=torch.ones_like(prompt), inputs_embeds=embeds, **kwargs)
        retro_ids = self.add_special_body_tokens(analysis, self.token_id_dict["<retro_body>"], self.num_body_tokens,
                                                 start_token_id=self.token_id_dict["<retro_start>"])
        hidden = self._query_hidden(retro_ids)
        cond = self.lm_to_graph_predictor(hidden.to(next(self.lm_to_graph_predictor.parameters()).dtype))
        reactants, scores, templates = self.graph_predictor.sample_templates(product, cond, product_smiles, topk)
        head = self.tokenizer.encode(f"To synthesize {product_smiles}, follow these procedures: ")
        return {"reactants": reactants, "scores": scores, "templates": templates,
                "analysis": head + analysis.cpu().squeeze().tolist()}

    @torch.no_grad()
    def one_step_reaction_batch(self, requests, topk, **kwargs):
        """``one_step_reaction`` for several products at once: one GIN-encoder forward over every spliced
        graph, ONE batched LLM decode over the left-padded prompts, one query-token forward, one predictor forward + top-k.
        requests: dicts with product_smiles, input_ids (1-D or None), design_text, molecule_graphs (or None).  Returns one
        result dict per request, as ``one_step_reaction`` would."""
        results: List[Optional[Dict[str, Any]]] = [None] * len(requests)
        prompts, graph_lists, products, live = [], [], [], []
        for i, rq in enumerate(requests):
            prompt = self.tokenizer.encode(f"{rq.get('design_text')} To synthesize <molecule>, follow these procedures: ",
                                           add_special_tokens=False, return_tensors="pt").to(self.device)
            ctx_ids, ctx_graphs = rq.get("input_ids"), rq.get("molecule_graphs")
            with_context = ctx_ids is not None and ctx_graphs is not None
            if with_context:
                prompt = torch.cat([ctx_ids.view(1, -1), prompt], dim=-1)
            product = self.smiles_to_graph(rq["product_smiles"])
            if product is None:
                results[i] = {"reactants": [], "scores": [], "templates": [],
                              "analysis": self.tokenizer.encode("Invalid product SMILES", add_special_tokens=False)}
                continue
            product.to(self.device)
            prompts.append(prompt)
            graph_lists.append((ctx_graphs.to_data_list() if with_context else []) + [product])
            products.append(product)
            live.append(i)
        shard = self._expansion_shard()
        if not live:
            # a rank whose replicated search diverged to zero requests must still meet its peers in the agreement check: they would
            # otherwise block in its all-reduce
            if shard is not None and hasattr(self.graph_predictor, "topk_templates_batch"):
                self._assert_ranks_agree(0, shard[2], "expansion requests of a lock-step round")
            return results
        smiles = [requests[i]["product_smiles"] for i in live]
        if shard is not None and hasattr(self.graph_predictor, "topk_templates_batch"):
            # Expansion-level split (second alternative; north_star: "RCCL all-gather of candidate scores"): the host A* is
            # replicated, every rank sees the same requests; rank r decodes and scores requests r, r + world, ..., then ONE all-gather of
            # the fixed-size records -- (topk_idx int32[k], topk_prob f32[k]) and the analysis tokens -- gives every rank every expansion;
            # template application and the merge (host) run on every rank for every request, so the trees stay identical.
            rank, world, group = shard
            self._assert_ranks_agree(len(live), group, "expansion requests of a lock-step round")
            mine = list(range(rank, len(live), world))
            cap = (len(live) + world - 1) // world
            nt = int(self.retro_max_new_tokens)
            idx = torch.zeros((cap, topk), dtype=torch.int32, device=self.device)
            prob = torch.zeros((cap, topk), dtype=torch.float32, device=self.device)
            toks = torch.full((cap, nt), -1, dtype=torch.int32, device=self.device)
            if mine:
                analysis, cond = self._decode_analysis([prompts[j] for j in mine], [graph_lists[j] for j in mine], kwargs)
                p_, i_ = self.graph_predictor.topk_templates_batch([products[j] for j in mine], cond, topk)
                k_ = min(topk, p_.shape[1])
                idx[:len(mine), :k_], prob[:len(mine), :k_] = i_[:, :k_].to(torch.int32), p_[:, :k_].float()
                toks[:len(mine), :min(nt, analysis.shape[1])] = analysis[:, :nt].to(torch.int32)
            from .distributed import all_gather_topk, gather_rows
            g_idx, g_prob = all_gather_topk(idx, prob, group=group)          # [world * cap, k], rank-major
            g_tok = gather_rows(toks, group=group)
            pos = [(j % world) * cap + j // world for j in range(len(live))]  # where request j of `live` sits in the gathered rows
            sel = torch.tensor(pos, dtype=torch.long, device=g_idx.device)
            probs_np, idx_np = g_prob[sel].float().cpu().numpy(), g_idx[sel].cpu().numpy()
            tok_rows = g_tok[sel].cpu()
            triples = self.graph_predictor.merge_topk(probs_np, idx_np, smiles)
            analyses = [[int(t) for t in row.tolist() if t >= 0] for row in tok_rows]
        else:
            analysis, cond = self._decode_analysis(prompts, graph_lists, kwargs)
            if hasattr(self.graph_predictor, "sample_templates_batch"):
                triples = self.graph_predictor.sample_templates_batch(products, cond, smiles, topk)
            else:
                triples = [self.graph_predictor.sample_templates(g, cond[j:j + 1], s, topk) for j, (g, s) in enumerate(zip(products, smiles))]
            analyses = [analysis[j].cpu().tolist() for j in range(len(live))]
        for j, i in enumerate(live):
            reactants, scores, templates = triples[j]
            head = self.tokenizer.encode(f"To synthesize {smiles[j]}, follow these procedures: ")
            results[i] = {"reactants": reactants, "scores": scores, "templates": templates, "analysis": head + analyses[j]}
        return results

    def _expansion_shard(self):
        """(rank, world, group) when a lock-step round's expansions / value estimates are split over the ranks (``expansion_shard``
        attribute, set by the driver: every rank then runs the SAME searches), else None."""
        sh = getattr(self, "expansion_shard", None)
        if sh is None:
            return None
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return None
        return sh

    def _assert_ranks_agree(self, n: int, group, what: str) -> None:
        """The expansion split issues collectives from inside a host A* that every rank replicates: it is only safe while all ranks take
        the same host decisions.  Before a gather whose buffer size depends on such a decision, check it (ONE 2-element all-reduce per
        round, next to an LLM decode): a disagreement raises on every rank instead of hanging some of them in RCCL."""
        import torch.distributed as dist
        backend = dist.get_backend(group)
        t = torch.tensor([n, -n], dtype=torch.int64, device=self.device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        if int(t[0]) != n or int(t[1]) != -n:
            raise RuntimeError(f"expansion_shard: the ranks disagree on the number of {what} ({n} here, {-int(t[1])}..{int(t[0])} elsewhere): "
                               "the replicated searches have diverged")

    def _shared_clock(self):
        """time.time() of rank 0, on every rank (expansion split only): the searches' max_planning_time is judged by ONE clock, so all ranks
        stop in the same round."""
        shard = self._expansion_shard()
        if shard is None:
            return time.time()
        import torch.distributed as dist
        _, _, group = shard
        backend = dist.get_backend(group)
        t = torch.tensor([time.time()], dtype=torch.float64, device=self.device if backend == "nccl" else "cpu")
        dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return float(t[0])

    def _decode_analysis(self, prompts, graph_lists, kwargs):
        """Device part of an expansion round for `prompts` (one per request): GIN-encode every spliced graph, ONE batched LLM decode over
        the left-padded prompts, one query-token forward -> (analysis tokens [n, new], predictor condition [n, 768])."""
        # ---- one encoder forward for every <molecule> slot of every prompt
        emb_layer = self.language_model.get_input_embeddings()
        all_graphs = GraphBatch.from_data_list([g for gl in graph_lists for g in gl])
        mol = self.graph_encoder(all_graphs.x, all_graphs.edge_index, all_graphs.edge_attr, all_graphs.batch)
        mol = self.graph_to_lm_connector(mol.to(next(self.graph_to_lm_connector.parameters()).dtype))
        L = max(p.shape[1] for p in prompts)
        embeds, mask, off = [], torch.zeros((len(prompts), L), dtype=torch.long, device=self.device), 0
        pad_id = self.tokenizer.eos_token_id
        for j, (p, gl) in enumerate(zip(prompts, graph_lists)):
            e = emb_layer(p)
            pos = (p == self.token_id_dict["<molecule>"]).nonzero()
            assert pos.shape[0] == len(gl), \
                f"Number of molecule tokens ({pos.shape[0]}) does not match number of molecule embeddings ({len(gl)})"
            e[pos[:, 0], pos[:, 1]] = mol[off:off + len(gl)].to(e.dtype)
            off += len(gl)
            if p.shape[1] < L:
                padding = emb_layer(torch.full((1, L - p.shape[1]), pad_id, dtype=torch.long, device=self.device))
                e = torch.cat([padding, e], dim=1)
            mask[j, L - p.shape[1]:] = 1
            embeds.append(e)
        embeds = torch.cat(embeds, dim=0)
        kwargs = dict(kwargs)
        if "max_new_tokens" in kwargs:
            kwargs["max_new_tokens"] = self.retro_max_new_tokens
        analysis = self._llm_generate(attention_mask=mask, inputs_embeds=embeds, **kwargs)
        retro_ids = self.add_special_body_tokens(analysis, self.token_id_dict["<retro_body>"], self.num_body_tokens,
                                                 start_token_id=self.token_id_dict["<retro_start>"])
        hidden = self._query_hidden(retro_ids)
        cond = self.lm_to_graph_predictor(hidden.to(next(self.lm_to_graph_predictor.parameters()).dtype))
        return analysis, cond

    _ANSWERS = ["All readily available", "Some commercial, some need 1-2 steps",
                "Mix of commercial and multi-step synthesis", "Mostly require complex synthesis",
                "All require extensive multi-step synthesis"]
    _ANSWER_COSTS = [0, 1, 2.5, 4.5, 7]

    def _complexity_prompt(self, smiles, reaction) -> str:
        if reaction is None:
            return f"""
                Estimate remaining steps for the target {smiles} consider the following factors::
                1. Intermediate complexity
                2. Reagent availability
                3. Side reactions
                4. Stereochemistry challenges"""
        reactants = ", ".join(r.mol for r in reaction.children)
        return f"""
                Estimate remaining steps for the target {smiles} given the following parameters:
                Current step {reaction.depth + 1},
                Current template: {reaction.template},
                Reactants: {reactants}. 
                Consider the following factors:
                1. Intermediate complexity
                2. Reagent availability
                3. Side reactions
                4. Stereochemistry challenges"""

    def _answer_tokens(self):
        return [self.tokenizer.encode(self.tokenizer.apply_chat_template(
            [{"role": "user", "content": "Estimate the synthesis complexity:"}, {"role": "assistant", "content": a}],
            tokenize=False, add_generation_prompt=False)) for a in self._ANSWERS]

    # The reference multiplies probs [5, 1] by a cost vector [5] (modeling_llamole.py:984-988): the product broadcasts to the 5 x 5
    # outer product, so its sum is sum(probs) * sum(costs) = 15 (up to f32 rounding) for EVERY molecule -- the language "cost" is a
    # constant.  That is what a user of the reference gets (pinned by tests/golden/host_traces.json), so it is the default here;
    # ``expected_cost_value = True`` switches to the evidently intended expectation sum_k p_k * cost_k.
    expected_cost_value = False
    # Opt-in consequence of the above (round 3): in the reference-compatible mode the language cost is sum(softmax) * sum(costs) = 15 for
    # every molecule -- the LLM forward behind it (one prefill per new tree node; ~100 nodes x ~130 tokens per expansion, 90 % of the
    # retrosynthesis workload's time on MI355X, bench.py --workload retro) only contributes f32 rounding noise of +-1e-6, which no two
    # implementations reproduce anyway.  True = return the constant without the forward.  Default False: the reference's structure, every
    # forward executed.  Ignored when ``expected_cost_value`` is set (then the logits matter).
    constant_language_cost_shortcut = False

    def _language_cost_is_constant(self) -> bool:
        return bool(self.constant_language_cost_shortcut) and not self.expected_cost_value

    def _cost_from_logits(self, logits, answer_tokens):
        """logits [n, vocab] of the last prompt position -> remaining-step cost per row (:976-993)."""
        answer_logits = torch.stack([logits[:, toks].mean(dim=1) for toks in answer_tokens])        # [5, n]
        probs = torch.softmax(answer_logits.float(), dim=0)
        costs = torch.tensor(self._ANSWER_COSTS, device=probs.device)
        if self.expected_cost_value:
            return (probs * costs[:, None]).sum(dim=0)                                              # [n]
        return torch.stack([(probs[:, j:j + 1] * costs).sum() for j in range(probs.shape[1])])      # the reference's outer product

    @torch.no_grad()
    def estimate_synthesis_complexity(self, smiles, input_ids=None, reaction=None, molecule_cost_weight=0,
                                      language_cost_weight=1, reference_tokens=None):
        cost = 0
        if molecule_cost_weight is not None and molecule_cost_weight > 0:
            cost += self.graph_predictor.estimate_cost(smiles) * molecule_cost_weight
        if language_cost_weight is not None and language_cost_weight > 0 and self._language_cost_is_constant():
            cost += float(sum(self._ANSWER_COSTS)) * language_cost_weight
        elif language_cost_weight is not None and language_cost_weight > 0:
            chat = self.tokenizer.apply_chat_template([{"role": "user", "content": self._complexity_prompt(smiles, reaction)}],
                                                      tokenize=False, add_generation_prompt=True)
            ids = self.tokenizer.encode(chat, return_tensors="pt").to(self.device)
            logits = self.language_model(ids).logits[:, -1, :]
            cost += self._cost_from_logits(logits, self._answer_tokens()).sum().item() * language_cost_weight
        return cost

    # prompts per LLM forward of estimate_synthesis_complexity_batch (env LLAMOLE_VALUE_BATCH overrides).  Measured on the retro workload
    # (bench.py --workload retro: Qwen2-7B, ~140-token prompts, ~1 300 prompts per call = the new tree nodes of all 16 lock-step searches
    # of a round): 256 -> 14.14 s per step, 512 -> 13.75 s, 1024 -> 13.52 s -- the prefill GEMMs of stock HF / hipBLASLt want M >= ~100 k
    # rows; 1024 prompts x 120 tokens x 18 944 x 2 B = 4.7 GB per MLP activation
    value_batch = 1024
    # Every value prompt opens with the same tokens (chat-template header + "Estimate remaining steps for the target"): their keys / values
    # are computed ONCE per call and every row's forward covers only its own remainder (causal attention: a token's state depends on
    # nothing after it, so each row still computes what its own full forward computes).  Used when the shared opening has at least this
    # many tokens; 0 / env LLAMOLE_VALUE_PREFIX=0 = every prompt forwarded whole.
    value_prefix_min = 8

    @staticmethod
    def _shared_opening(rows) -> int:
        """Number of leading tokens common to all rows, leaving every row at least one token of its own."""
        lo, hi = min(rows), max(rows)           # lexicographic extremes: their common prefix is the common prefix of all rows
        p = 0
        for a, b in zip(lo, hi):
            if a != b:
                break
            p += 1
        return max(0, min(p, min(len(r) for r in rows) - 1))

    def _opening_cache(self, opening):
        """Per-layer (keys, values) of ONE forward over the shared opening tokens, or None when the LM has no KV-cache interface."""
        ids = torch.tensor([opening], dtype=torch.long, device=self.device)
        try:
            out = self.language_model(input_ids=ids, attention_mask=torch.ones_like(ids), use_cache=True, logits_to_keep=1)
        except TypeError:
            return None
        pkv = getattr(out, "past_key_values", None)
        layers = getattr(pkv, "layers", None)
        if not layers or any(getattr(l, "keys", None) is None or getattr(l, "sliding_window", None) for l in layers):
            return None
        return [(l.keys, l.values) for l in layers]

    @torch.no_grad()
    def estimate_synthesis_complexity_batch(self, items, input_ids=None, molecule_cost_weight=0, language_cost_weight=1,
                                            max_batch: Optional[int] = None) -> List[float]:
        """``estimate_synthesis_complexity`` for many ``(smiles, reaction)`` pairs with ONE LLM forward per ``max_batch``
        prompts instead of one per new tree node: prompts are left-padded, padding is masked and position
        ids count real tokens only, so every row computes exactly what its own unpadded forward computes; the tokens all
        prompts open with go through the LLM once per call (``value_prefix_min``)."""
        n = len(items)
        costs = [0.0] * n
        shard = self._expansion_shard()
        if n == 0:
            if shard is not None and not getattr(self, "_in_value_shard", False) and not self._language_cost_is_constant():
                self._assert_ranks_agree(0, shard[2], "value prompts of a lock-step round")      # see expand_batch: no early return past the check
            return costs
        if shard is not None and not getattr(self, "_in_value_shard", False) and not self._language_cost_is_constant():
            # the round's value prompts split over the ranks (item i on rank i % world), ONE all-gather of the float64 costs
            rank, world, group = shard
            self._assert_ranks_agree(n, group, "value prompts of a lock-step round")
            mine = list(range(rank, n, world))
            cap = (n + world - 1) // world
            self._in_value_shard = True
            try:
                local = self.estimate_synthesis_complexity_batch([items[i] for i in mine], input_ids, molecule_cost_weight,
                                                                 language_cost_weight, max_batch) if mine else []
            finally:
                self._in_value_shard = False
            buf = torch.zeros((cap, 1), dtype=torch.float64, device=self.device)
            if local:
                buf[:len(local), 0] = torch.tensor(local, dtype=torch.float64)
            from .distributed import gather_rows
            allc = gather_rows(buf, group=group).cpu()
            return [float(allc[(i % world) * cap + i // world, 0]) for i in range(n)]
        if max_batch is None:
            max_batch = int(os.environ.get("LLAMOLE_VALUE_BATCH", self.value_batch))
        if molecule_cost_weight is not None and molecule_cost_weight > 0:
            for i, (smiles, _) in enumerate(items):
                costs[i] += self.graph_predictor.estimate_cost(smiles) * molecule_cost_weight
        if language_cost_weight is not None and language_cost_weight > 0 and self._language_cost_is_constant():
            const = float(sum(self._ANSWER_COSTS)) * language_cost_weight
            return [c + const for c in costs]
        if language_cost_weight is not None and language_cost_weight > 0:
            answer_tokens = self._answer_tokens()
            rows = [self.tokenizer.encode(self.tokenizer.apply_chat_template(
                [{"role": "user", "content": self._complexity_prompt(smiles, reaction)}], tokenize=False,
                add_generation_prompt=True)) for smiles, reaction in items]
            pad = getattr(self.tokenizer, "pad_token_id", None)
            pad = self.tokenizer.eos_token_id if pad is None else pad
            pmin = int(os.environ.get("LLAMOLE_VALUE_PREFIX", self.value_prefix_min))
            P = self._shared_opening(rows) if (pmin > 0 and n > 1) else 0
            opening = self._opening_cache(rows[0][:P]) if P >= max(pmin, 1) else None
            if opening is None:
                P = 0
            self.last_value_opening = P           # tokens per prompt served from the shared keys / values (0 = whole prompts forwarded)
            self.value_tokens_forwarded = getattr(self, "value_tokens_forwarded", 0) + sum(len(r) for r in rows) - (n - 1) * P
            # prompts of similar length share a forward (less padding; every row still computes exactly its own unpadded forward), and
            # the host never waits between forwards: the per-chunk costs stay on the device until all chunks are enqueued
            order = sorted(range(n), key=lambda i: len(rows[i]))
            # chunks are capped by TOKENS as well as by prompts: with the shared opening every row of a chunk keeps keys / values for
            # P + L positions in every layer during its forward (value_batch was tuned on ~140-token prompts of Qwen2-7B: ~8 GB; longer
            # prompts or more KV heads must not grow that without bound).  LLAMOLE_VALUE_TOKENS / value_token_cap overrides.
            tok_cap = int(os.environ.get("LLAMOLE_VALUE_TOKENS", getattr(self, "value_token_cap", 160 * 1024)))
            chunks, lo = [], 0
            while lo < n:
                hi = lo + 1
                while hi < n and hi - lo < max_batch and (hi - lo + 1) * len(rows[order[hi]]) <= tok_cap:
                    hi += 1
                chunks.append(order[lo:hi])
                lo = hi
            pending = []
            for sel in chunks:
                chunk = [rows[i][P:] for i in sel]
                L = max(len(r) for r in chunk)
                ids = torch.full((len(chunk), L), pad, dtype=torch.long)
                mask = torch.zeros((len(chunk), P + L), dtype=torch.long)
                mask[:, :P] = 1
                for j, r in enumerate(chunk):
                    ids[j, L - len(r):] = torch.tensor(r, dtype=torch.long)
                    mask[j, P + L - len(r):] = 1
                ids, mask = ids.to(self.device, non_blocking=True), mask.to(self.device, non_blocking=True)
                posid = (mask.cumsum(dim=1) - 1).clamp_min(0)[:, P:]
                kw = dict(input_ids=ids, attention_mask=mask, position_ids=posid)
                if opening is not None:      # [opening keys/values | padding | the row's own tokens]: padding masked, positions count real tokens
                    from transformers import DynamicCache
                    B = len(chunk)
                    cache = DynamicCache(ddp_cache_data=[(k.expand(B, -1, -1, -1), v.expand(B, -1, -1, -1)) for k, v in opening])
                    try:
                        logits = self.language_model(logits_to_keep=1, use_cache=True, past_key_values=cache, **kw).logits[:, -1, :]
                    except torch.OutOfMemoryError:
                        # the cached form keeps every layer's keys / values of the chunk resident: forward the WHOLE prompts without a
                        # cache instead (same costs; what the path did before the shared opening)
                        del cache
                        torch.cuda.empty_cache()
                        full = [rows[i] for i in sel]
                        Lf = max(len(r) for r in full)
                        idf = torch.full((len(full), Lf), pad, dtype=torch.long)
                        mf = torch.zeros((len(full), Lf), dtype=torch.long)
                        for j, r in enumerate(full):
                            idf[j, Lf - len(r):] = torch.tensor(r, dtype=torch.long)
                            mf[j, Lf - len(r):] = 1
                        idf, mf = idf.to(self.device), mf.to(self.device)
                        logits = self.language_model(input_ids=idf, attention_mask=mf, position_ids=(mf.cumsum(dim=1) - 1).clamp_min(0),
                                                     logits_to_keep=1, use_cache=False).logits[:, -1, :]
                        cache = None
                    del cache
                else:
                    try:      # only the last position's logits are read: no [B, L, vocab] product, no KV cache
                        logits = self.language_model(logits_to_keep=1, use_cache=False, **kw).logits[:, -1, :]
                    except TypeError:
                        logits = self.language_model(**kw).logits[:, -1, :]
                pending.append((sel, self._cost_from_logits(logits, answer_tokens) * language_cost_weight))
            for sel, c in pending:
                for i, v in zip(sel, c.tolist()):
                    costs[i] += v
        return costs

    def _create_failure_result(self, target_smiles, generated_tokens=None) -> Dict[str, Any]:
        return {"target": target_smiles, "success": False, "time": 0.0, "reaction_list": None, "cost": None,
                "templates": None, "route_length": None,
                "analysis_tokens": generated_tokens if generated_tokens is not None else "<NO ANALYSIS>"}

    @torch.no_grad()
    def retrosynthesize(self, input_ids, smiles=None, molecule_graphs=None, expansion_topk=50, iterations=100,
                        starting_mols=None, molecule_cost_weight=0, language_cost_weight=1, max_planning_time=300,
                        rollback=True, design_text=None, **kwargs) -> Dict[str, Any]:
        if starting_mols is None:
            if self.graph_predictor.available is None:
                raise ValueError("No starting molecules provided and no available starting molecules found.")
            starting_mols = self.graph_predictor.available["smiles"].tolist()
        if smiles is None and rollback:
            return self._create_failure_result(None, self.retrosynthesize_rollback(input_ids, design_text, None, **kwargs))
        target = smiles.replace("*", "[H]") if "*" in smiles else smiles
        if not self.graph_decoder.check_valid(target) and rollback:
            return self._create_failure_result(target, self.retrosynthesize_rollback(input_ids, design_text, target, **kwargs))
        t0 = time.time()
        known = starting_mols if isinstance(starting_mols, (set, frozenset)) else set(starting_mols)
        success, route, _ = molstar(
            target_mol=target, target_mol_id=0, starting_mols=known,
            expand_fn=lambda s: self.one_step_reaction(s, input_ids=input_ids, design_text=design_text,
                                                       molecule_graphs=molecule_graphs, topk=expansion_topk, **kwargs),
            value_fn=lambda s, r: self.estimate_synthesis_complexity(s, input_ids, r, molecule_cost_weight, language_cost_weight),
            iterations=iterations, max_time=max_planning_time,
            value_batch_fn=(lambda items: self.estimate_synthesis_complexity_batch(items, input_ids, molecule_cost_weight,
                                                                                     language_cost_weight))
            if self.batch_values else None)
        total = time.time() - t0
        if success:
            reactions, templates, cost, analysis = route.get_reaction_list()
            return {"target": target, "success": True, "time": total, "reaction_list": reactions, "cost": cost,
                    "templates": templates, "analysis_tokens": analysis, "route_length": route.length}
        if rollback:
            return self._create_failure_result(target, self.retrosynthesize_rollback(input_ids, design_text, target, **kwargs))
        return {"target": target, "success": False, "time": total, "reaction_list": None, "cost": None,
                "templates": None, "analysis_tokens": None, "route_length": None}

    @torch.no_grad()
    def retrosynthesize_many(self, input_ids_list, smiles_list, molecule_graphs=None, expansion_topk=50, iterations=100,
                             starting_mols=None, molecule_cost_weight=0, language_cost_weight=1, max_planning_time=300,
                             rollback=True, design_text=None, **kwargs) -> List[Dict[str, Any]]:
        """``retrosynthesize`` for several targets with their A* searches advanced in lock step (planner.molstar_many):
        each round's expansions share one batched LLM decode / GIN forward (one_step_reaction_batch) and each expansion's
        new nodes share one value forward.  Pre-checks, rollbacks and result dicts are those of ``retrosynthesize``;
        ``max_planning_time`` is measured on the shared clock."""
        from .planner import molstar_many
        if starting_mols is None:
            if self.graph_predictor.available is None:
                raise ValueError("No starting molecules provided and no available starting molecules found.")
            starting_mols = self.graph_predictor.available["smiles"].tolist()
        known = starting_mols if isinstance(starting_mols, (set, frozenset)) else set(starting_mols)
        out: List[Optional[Dict[str, Any]]] = [None] * len(smiles_list)
        targets, where = [], []
        for i, smiles in enumerate(smiles_list):
            ids = input_ids_list[i]
            if smiles is None and rollback:
                out[i] = self._create_failure_result(None, self.retrosynthesize_rollback(ids, design_text, None, **kwargs))
                continue
            target = smiles.replace("*", "[H]") if "*" in smiles else smiles
            if not self.graph_decoder.check_valid(target) and rollback:
                out[i] = self._create_failure_result(target, self.retrosynthesize_rollback(ids, design_text, target, **kwargs))
                continue
            targets.append(target)
            where.append(i)
        if targets:
            t0 = time.time()

            def expand_batch(picks):
                reqs = [dict(product_smiles=mol, input_ids=input_ids_list[where[k]], design_text=design_text,
                             molecule_graphs=molecule_graphs) for k, mol in picks]
                return self.one_step_reaction_batch(reqs, expansion_topk, **kwargs)
            outcomes = molstar_many(
                targets, known, expand_batch,
                value_fn=lambda s, r: self.estimate_synthesis_complexity(s, None, r, molecule_cost_weight, language_cost_weight),
                iterations=iterations, max_time=max_planning_time,
                value_batch_fn=lambda items: self.estimate_synthesis_complexity_batch(items, None, molecule_cost_weight,
                                                                                      language_cost_weight),
                clock=self._shared_clock if self._expansion_shard() is not None else None)
            total = time.time() - t0
            for k, (success, route, _) in enumerate(outcomes):
                i, target = where[k], targets[k]
                if success:
                    reactions, templates, cost, analysis = route.get_reaction_list()
                    out[i] = {"target": target, "success": True, "time": total, "reaction_list": reactions, "cost": cost,
                              "templates": templates, "analysis_tokens": analysis, "route_length": route.length}
                elif rollback:
                    out[i] = self._create_failure_result(target, self.retrosynthesize_rollback(input_ids_list[i], design_text,
                                                                                                  target, **kwargs))
                else:
                    out[i] = {"target": target, "success": False, "time": total, "reaction_list": None, "cost": None,
                              "templates": None, "analysis_tokens": None, "route_length": None}
        return out

    # ------------------------------------------------------------------ top level
    @torch.no_grad()
    def generate(self, input_ids=None, attention_mask=None, molecule_properties=None, molecule_graphs=None,
                 rollback=False, starting_mols=None, expansion_topk=50, iterations=100, molecule_cost_weight=0,
                 language_cost_weight=1, do_molecular_design=True, do_retrosynthesis=True, input_smiles_list=None,
                 max_planning_time=30, design_text_list=None, **kwargs) -> Dict:
        if attention_mask is None:
            attention_mask = input_ids.new_ones(input_ids.shape)
        info: Dict[str, Any] = {"token_lists": [], "text_lists": [], "design_analysis_tokens": None,
                                "smiles_list": None, "retro_plan_dict": None}
        if do_molecular_design is True:
            tokens, smiles = self.design_molecule(input_ids, attention_mask, molecule_properties, molecule_graphs,
                                                  rollback, **kwargs)
            info["design_analysis_tokens"] = tokens.cpu()
            info["smiles_list"] = smiles
        elif input_smiles_list is not None:
            info["smiles_list"] = input_smiles_list
        else:
            raise ValueError("Either do_molecular_design must be True/False or input_smiles_list must be provided.")
        if do_retrosynthesis:
            info["retro_plan_dict"] = {}
            design_text = design_text_list[0] if design_text_list is not None else None
            if self.batch_retro and len(info["smiles_list"]) > 1:
                ids_list = [input_ids[i] if input_ids.dim() > 1 else input_ids for i in range(len(info["smiles_list"]))]
                plans = self.retrosynthesize_many(
                    ids_list, info["smiles_list"], molecule_graphs=molecule_graphs, starting_mols=starting_mols,
                    expansion_topk=expansion_topk, iterations=iterations, molecule_cost_weight=molecule_cost_weight,
                    language_cost_weight=language_cost_weight, max_planning_time=max_planning_time, design_text=design_text,
                    **kwargs)
                for smiles, plan in zip(info["smiles_list"], plans):
                    info["retro_plan_dict"][smiles] = plan
            for i, smiles in enumerate(info["smiles_list"] if not (self.batch_retro and len(info["smiles_list"]) > 1) else []):
                info["retro_plan_dict"][smiles] = self.retrosynthesize(
                    input_ids[i] if input_ids.dim() > 1 else input_ids, smiles, molecule_graphs=molecule_graphs,
                    starting_mols=starting_mols, expansion_topk=expansion_topk, iterations=iterations,
                    molecule_cost_weight=molecule_cost_weight, language_cost_weight=language_cost_weight,
                    max_planning_time=max_planning_time, design_text=design_text, **kwargs)
        else:
            info["retro_plan_dict"] = {s: {"success": None} for s in info["smiles_list"]}
        available = None
        for b, mol in enumerate(info["smiles_list"]):
            tokens: List[int] = []
            texts: List[str] = []
            ignore: Dict[int, Any] = {}
            if do_molecular_design:
                dt = info["design_analysis_tokens"][b].tolist()
                tokens = dt + [IGNORE_INDEX]
                if mol is None:
                    mol = "<NO MOLECULE>"
                texts = [self.tokenizer.decode(dt, skip_special_tokens=True), mol + ". "]
                ignore = {0: mol}
            if do_retrosynthesis:
                if available is None:
                    available = set(self.graph_predictor.available["smiles"].tolist())
                plan = info["retro_plan_dict"][mol]
                if plan["success"]:
                    for reaction, template, cost, at in zip(plan["reaction_list"], plan["templates"], plan["cost"], plan["analysis_tokens"]):
                        at = at.tolist() if isinstance(at, torch.Tensor) else at
                        tokens.extend(at + [IGNORE_INDEX])
                        texts.extend([self.tokenizer.decode(at, skip_special_tokens=True),
                                      reaction if reaction is not None else "<NO REACTION>", " with the template ",
                                      template if template is not None else "<NO TEMPLATE>", " which requires the reactants: "])
                        if reaction is not None:
                            rs = reaction.split(">>")[1].split(".")
                            texts.extend([", ".join(f"{r} (available)" if r in available else r for r in rs), ". "])
                        else:
                            texts.extend(["<NO REACTANTS>. "])
                        ignore[len(tokens) - 1] = (reaction, template, cost)
                else:
                    at = plan["analysis_tokens"]
                    at = at.tolist() if isinstance(at, torch.Tensor) else at
                    tokens.extend(at)
                    texts.extend([self.tokenizer.decode(at, skip_special_tokens=True), " <NO REACTION FOUND>"])
            info["token_lists"].append(tokens)
            info["text_lists"].append(texts)
            info[f"batch_{b}_ignore_positions"] = ignore
        info["IGNORE_INDEX"] = IGNORE_INDEX
        try:      # a malformed graph batch of the LAST GIN call of this run must not go unreported (the flag is otherwise found by the next call)
            from .graph_encoder import check_graph_errors
            check_graph_errors(wait=True)
        except ImportError:
            pass
        return info
