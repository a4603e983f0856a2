#!/usr/bin/env python3
"""Side-by-side pin of the host chemistry tails against the REFERENCE's own code, with the real rdkit / rdchiral (tools/chem_pin.sh runs
this inside a virtualenv that has them; it cannot run in the build image or on the GPU box, where neither package exists).

    python tools/chem_pin_compare.py --reference /path/to/Llamole [--graphs 200] [--template-cases 40]

1. graph -> SMILES: ``graph_decoder/molecule_utils.graph_to_smiles`` of the reference (molecule_utils.py:49-111, with
   build_molecule_with_partial_charges :113-166, correct_mol :169-210, connect_fragments, mol2smiles, check_polymer :322-352) and
   ``llamole_amd.molecule_utils.graph_to_smiles`` on the same seeded integer graphs -- random trees with ring closures, over-valent atoms,
   disconnected fragments, polymer stars, single atoms, aromatic rings: equal lists of Optional[str], element by element.
2. template application + merge (a17 / f3): the reference's ``GraphPredictor.sample_templates`` (graph_predictor/model.py:164-228) driven
   by a stand-in predictor that returns scripted logits, against ``llamole_amd.graph_predictor.merge_template_outcomes`` fed the same
   top-k (probabilities, templates): equal reactant lists, equal template lists, scores equal to 1e-12 -- with rdchiral applying real
   retro templates to real products.
3. SMILES -> integer graph (a18): the reference's ``smiles_to_graph`` body (modeling_llamole.py:720-760) is a method of a class that
   cannot be imported without peft / trl; its arithmetic is restated inline below from those lines and compared with this repository's.

Only torch_geometric is stubbed (the reference's graph_predictor imports its MessagePassing base class; the stand-in predictor never
runs it).  Exit code 0 = every comparison equal.
"""
import argparse
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ATOM_DECODER = ["C", "N", "O", "F", "S", "Cl", "Br", "P", "*"]
SINGLE, DOUBLE, TRIPLE, AROMATIC = 1, 2, 3, 4


def seeded_graphs(n, seed=0):
    """Integer graphs as GraphDiT hands them to graph_to_smiles: (atom_types [k] int64, edge_types [k, k] int64 symmetric, 0 = no bond)."""
    rng = np.random.default_rng(seed)
    out = []
    for g in range(n):
        kind = g % 8
        k = int(rng.integers(1, 3)) if kind == 7 else int(rng.integers(3, 28))
        atoms = rng.choice(len(ATOM_DECODER) - 1, size=k, p=[0.55, 0.12, 0.14, 0.04, 0.05, 0.04, 0.03, 0.03])
        e = np.zeros((k, k), dtype=np.int64)
        for i in range(1, k):                                   # a random tree ...
            if kind == 3 and rng.random() < 0.15:
                continue                                        # ... with missing edges: disconnected fragments
            j = int(rng.integers(0, i))
            e[i, j] = e[j, i] = int(rng.choice([SINGLE, SINGLE, SINGLE, DOUBLE, TRIPLE] if kind != 1 else [SINGLE]))
        for _ in range(int(rng.integers(0, 3))):                # ring closures
            i, j = (int(v) for v in rng.integers(0, k, 2))
            if i != j and e[i, j] == 0:
                e[i, j] = e[j, i] = SINGLE
        if kind == 2 and k >= 6:                                # an aromatic six-ring on the first six atoms
            e[:6, :6] = 0
            atoms[:6] = 0
            for i in range(6):
                e[i, (i + 1) % 6] = e[(i + 1) % 6, i] = AROMATIC
            for i in range(6, k):
                if e[i].sum() == 0:
                    j = int(rng.integers(0, 6))
                    e[i, j] = e[j, i] = SINGLE
        if kind == 4:                                           # over-valent hetero atoms (formal-charge branch)
            c = int(rng.integers(0, k))
            atoms[c] = int(rng.choice([1, 2, 4]))
            for j in rng.permutation(k)[:5]:
                if j != c:
                    e[c, j] = e[j, c] = SINGLE
        if kind == 5:                                           # over-valent carbon (bond-order reduction)
            c = int(rng.integers(0, k))
            atoms[c] = 0
            for j in rng.permutation(k)[:3]:
                if j != c:
                    e[c, j] = e[j, c] = int(rng.choice([DOUBLE, TRIPLE]))
        if kind == 6:                                           # polymer stars
            for c in rng.permutation(k)[:int(rng.integers(1, 3))]:
                atoms[c] = len(ATOM_DECODER) - 1
        out.append((torch.from_numpy(atoms.astype(np.int64)), torch.from_numpy(e)))
    return out


def stub_torch_geometric():
    import torch.nn as nn
    tg, tgn, tgu = types.ModuleType("torch_geometric"), types.ModuleType("torch_geometric.nn"), types.ModuleType("torch_geometric.utils")

    class MessagePassing(nn.Module):
        def __init__(self, aggr="add"):
            super().__init__()

    tgn.MessagePassing = MessagePassing
    for name in ("global_add_pool", "global_max_pool", "global_mean_pool"):
        setattr(tgn, name, lambda *a, **k: (_ for _ in ()).throw(RuntimeError("not used by chem_pin_compare")))
    for name in ("to_dense_adj", "to_dense_batch", "remove_self_loops"):
        setattr(tgu, name, lambda *a, **k: (_ for _ in ()).throw(RuntimeError("not used by chem_pin_compare")))
    tg.nn, tg.utils = tgn, tgu
    for k, v in (("torch_geometric", tg), ("torch_geometric.nn", tgn), ("torch_geometric.utils", tgu)):
        sys.modules.setdefault(k, v)


# a few retro templates of the USPTO kind (amide / ester / ether / Suzuki-like disconnections) and products they apply to
TEMPLATES = [
    "[C:1](=[O:2])-[N:3]>>[C:1](=[O:2])-O.[N:3]",
    "[C:1](=[O:2])-[O:3]-[C:4]>>[C:1](=[O:2])-O.[O:3]-[C:4]",
    "[c:1]-[c:2]>>[c:1]-Br.[c:2]-B(O)O",
    "[C:1]-[O:2]-[C:3]>>[C:1]-O.[C:3]-[O:2]",
    "[C:1]-[N:2]>>[C:1]-Br.[N:2]",
    "[N:1]-[S:2](=[O:3])=[O:4]>>[N:1].Cl-[S:2](=[O:3])=[O:4]",
    "[C:1]=[C:2]>>[C:1]-Br.[C:2]",
    "[O:1]-[C:2]>>[O:1].[C:2]-I",
]
PRODUCTS = ["CC(=O)NCc1ccccc1", "CCOC(=O)c1ccc(cc1)-c1ccccc1", "COc1ccc(CNC(C)=O)cc1", "CCN(CC)S(=O)(=O)c1ccccc1", "CC(=O)OCC=C", "c1ccc(cc1)-c1ccncc1",
            "CC(C)OC(=O)CNC(=O)C", "O=C(Nc1ccccc1)c1ccco1"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", required=True, help="checkout of liugangcode/Llamole")
    ap.add_argument("--graphs", type=int, default=200)
    ap.add_argument("--template-cases", type=int, default=40)
    args = ap.parse_args()
    import rdkit                                           # noqa: F401 -- the point of this script
    from rdchiral.main import rdchiralRunText              # noqa: F401
    stub_torch_geometric()
    sys.path.insert(0, os.path.join(args.reference, "src", "model"))
    import graph_decoder.molecule_utils as ref_mu          # the reference, imported, not copied
    import graph_predictor.model as ref_gp
    from llamole_amd import molecule_utils as our_mu
    from llamole_amd.graph_predictor import merge_template_outcomes

    # ---- 1. graph -> SMILES
    graphs = seeded_graphs(args.graphs)
    want = ref_mu.graph_to_smiles([(a.clone(), e.clone()) for a, e in graphs], ATOM_DECODER)
    got = our_mu.graph_to_smiles([(a.clone(), e.clone()) for a, e in graphs], ATOM_DECODER)
    bad = [(i, w, g) for i, (w, g) in enumerate(zip(want, got)) if w != g]
    print(f"graph_to_smiles: {len(graphs)} graphs, {sum(w is not None for w in want)} valid in the reference, {len(bad)} differ")
    for i, w, g in bad[:10]:
        print(f"   graph {i}: reference {w!r}  ours {g!r}")
    for s in {w for w in want if w}:
        assert ref_mu.check_valid(s) == our_mu.check_valid(s), s
    ok = not bad

    # ---- 2. template application + merge: the reference's sample_templates with scripted logits
    rng = np.random.default_rng(1)
    nt = len(TEMPLATES)

    class ScriptedPredictor(torch.nn.Module):
        logits = None

        def forward(self, x, edge_index, edge_attr, batch, c):
            return self.logits.clone()

    gp = ref_gp.GraphPredictor.__new__(ref_gp.GraphPredictor)
    torch.nn.Module.__init__(gp)
    gp.predictor = ScriptedPredictor()
    gp.text_drop = 0.0
    gp.label_to_template = {i: t for i, t in enumerate(TEMPLATES)}
    pg = types.SimpleNamespace(x=torch.zeros(3, dtype=torch.long), edge_index=torch.zeros(2, 0, dtype=torch.long), edge_attr=torch.zeros(0, dtype=torch.long))
    n_bad = 0
    for case in range(args.template_cases):
        product = PRODUCTS[case % len(PRODUCTS)]
        logits = torch.from_numpy(rng.standard_normal((1, nt)).astype(np.float32) * 2)
        gp.predictor.logits = logits
        k = int(rng.integers(2, nt + 1))
        r_ref, s_ref, t_ref = gp.sample_templates(pg, None, product, topk=k)
        p, idx = torch.topk(torch.softmax(logits, dim=1), k=k, dim=1)
        r_our, s_our, t_our = merge_template_outcomes(p.float().numpy()[0], [TEMPLATES[int(i)] for i in idx[0]], product, rdchiralRunText)
        same = list(r_ref) == list(r_our) and list(t_ref) == list(t_our) and np.allclose(s_ref, s_our, rtol=0, atol=1e-12)
        if not same:
            n_bad += 1
            print(f"   sample_templates case {case} ({product}, top-{k}): reference {r_ref} {s_ref}  ours {r_our} {s_our}")
    print(f"sample_templates merge: {args.template_cases} cases, {n_bad} differ")
    ok = ok and n_bad == 0

    # ---- 3. SMILES -> integer graph (modeling_llamole.py:720-760, restated inline from those lines: the class needs peft / trl to import)
    from rdkit import Chem
    from llamole_amd.modeling_llamole import GraphLLMForCausalMLM
    bond_index = {"SINGLE": 1, "DOUBLE": 2, "TRIPLE": 3, "AROMATIC": 4}
    n_bad = 0
    for smi in [w for w in want if w][:100] + PRODUCTS:
        mol = Chem.MolFromSmiles(smi)
        if mol is None:
            continue
        x_ref = [117 if a.GetSymbol() == "*" else a.GetAtomicNum() - 2 for a in mol.GetAtoms() if a.GetAtomicNum() != 1]
        src, dst, typ = [], [], []
        for b in mol.GetBonds():
            i, j = b.GetBeginAtomIdx(), b.GetEndAtomIdx()
            if mol.GetAtomWithIdx(i).GetAtomicNum() != 1 and mol.GetAtomWithIdx(j).GetAtomicNum() != 1:
                src += [i, j]
                dst += [j, i]
                typ += [bond_index.get(str(b.GetBondType()), 1)] * 2
        g = GraphLLMForCausalMLM.smiles_to_graph(None, smi)
        same = g is not None and g.x.tolist() == x_ref and g.edge_index.tolist() == ([src, dst] if src else [[], []]) and g.edge_attr.tolist() == typ
        n_bad += not same
    print(f"smiles_to_graph: {n_bad} differ")
    ok = ok and n_bad == 0
    print("chem_pin_compare:", "EQUAL -- rows a15 / f3 pinned" if ok else "DIFFERENCES FOUND")
    return 0 if ok else 1


if __name__ == "__main__":
    raise SystemExit(main())
