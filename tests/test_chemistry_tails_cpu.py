"""The host chemistry tails EXECUTED once.

rdkit / rdchiral exist in neither image, so `llamole_amd/molecule_utils.py` (reference graph_decoder/molecule_utils.py:49-352) and the
rdkit / rdchiral branches of `graph_predictor.py` / `modeling_llamole.smiles_to_graph` had only ever been read, never run.  These tests
drive every branch of them through `tests/fake_rdkit.py` -- the over-valence -> +1 formal charge repair, the bond-order reduction, the
aromatic dead end, fragment connection (success and failure), largest-fragment selection, the polymer check, the exception fallback, the
Morgan fingerprint, the default template runner, SMILES -> integer graph.  The double is NOT RDKit: these tests pin nothing about
chemistry (a15 / f3 stay partial); they end "a typo in correct_mol's loop would ship".
"""
import numpy as np
import pytest
import torch

from tests import fake_rdkit

DEC = ["C", "N", "O", "F", "S", "Cl", "*"]       # atom_decoder: class index -> element symbol


def _graph(atoms, bonds):
    n = len(atoms)
    e = torch.zeros(n, n, dtype=torch.long)
    for i, j, t in bonds:
        e[i, j] = e[j, i] = t
    return torch.tensor([DEC.index(a) for a in atoms]), e


@pytest.fixture
def chem(monkeypatch):
    return fake_rdkit.install(monkeypatch, template_outcomes=lambda t, s: {"T_two": [s + "_b.a_" + s], "T_none": [], "T_boom": None}[t]
                              if t != "T_raise" else (_ for _ in ()).throw(RuntimeError("template failed")))


def test_build_molecule_adds_formal_charge_to_over_valent_n_o_s(chem):
    from llamole_amd import molecule_utils as mu
    atoms, e = _graph(["N", "C", "C", "C", "C"], [(0, 1, 1), (0, 2, 1), (0, 3, 1), (0, 4, 1)])      # N with four single bonds
    mol = mu.build_molecule(atoms, e, DEC)
    assert mol.GetAtomWithIdx(0).GetFormalCharge() == 1 and mu.valency_problem(mol) == (True, None)
    atoms, e = _graph(["O", "C", "C", "C"], [(0, 1, 1), (0, 2, 1), (0, 3, 1)])                         # O+ with three
    assert mu.build_molecule(atoms, e, DEC).GetAtomWithIdx(0).GetFormalCharge() == 1
    atoms, e = _graph(["C", "C", "C", "C", "C", "C"], [(0, 1, 1), (0, 2, 1), (0, 3, 1), (0, 4, 1), (0, 5, 1)])   # carbon is never charged
    mol = mu.build_molecule(atoms, e, DEC)
    assert mol.GetAtomWithIdx(0).GetFormalCharge() == 0
    ok, info = mu.valency_problem(mol)
    assert not ok and info == [0, 5]


def test_correct_mol_lowers_the_highest_bond_order_until_legal(chem):
    from llamole_amd import molecule_utils as mu
    atoms, e = _graph(["C", "C", "O", "C"], [(0, 1, 2), (0, 2, 2), (0, 3, 1)])      # C with valence 5: one double bond must become single
    mol = mu.build_molecule(atoms, e, DEC)
    fixed, was_ok = mu.correct_mol(mol, connection=False)
    assert fixed is not None and was_ok is False and mu.valency_problem(fixed)[0]
    orders = sorted(int(b.GetBondType()) for b in fixed.GetAtomWithIdx(0).GetBonds())
    assert orders == [1, 1, 2]
    ok_mol = mu.build_molecule(*_graph(["C", "O"], [(0, 1, 2)]), DEC)
    same, was_ok = mu.correct_mol(ok_mol)
    assert same is ok_mol and was_ok is True
    # valence 5 from single bonds only: a single bond is REMOVED (order 1 - 1 = 0 -> no bond added back)
    mol = mu.build_molecule(*_graph(["C"] * 6, [(0, 1, 1), (0, 2, 1), (0, 3, 1), (0, 4, 1), (0, 5, 1)]), DEC)
    fixed, _ = mu.correct_mol(mol)
    assert fixed is not None and len(fixed.GetAtomWithIdx(0).GetBonds()) == 4


def test_correct_mol_gives_up_on_all_aromatic_over_valence(chem):
    from llamole_amd import molecule_utils as mu
    mol = mu.build_molecule(*_graph(["O", "C", "C"], [(0, 1, 4), (0, 2, 4)]), DEC)      # O with two aromatic bonds = 3 > 2, nothing to lower
    mol.GetAtomWithIdx(0).SetFormalCharge(0)
    assert mu.valency_problem(mol)[0] is False
    assert mu.correct_mol(mol) == (None, False)


def test_connect_fragments_joins_by_free_valence_or_fails(chem):
    from llamole_amd import molecule_utils as mu
    mol = mu.build_molecule(*_graph(["C", "C", "O", "N"], [(0, 1, 1)]), DEC)            # C-C, O, N: three fragments
    joined = mu.connect_fragments(mol)
    assert joined is not None and len(chem.GetMolFrags(joined)) == 1 and joined.GetNumAtoms() == 4
    assert mu.connect_fragments(joined) is joined                                        # a connected molecule is returned as it is
    sat = mu.build_molecule(*_graph(["F", "F", "Cl", "Cl"], [(0, 1, 1), (2, 3, 1)]), DEC)   # F2 + Cl2: no atom has free valence
    assert mu.connect_fragments(sat) is None
    fixed, _ = mu.correct_mol(sat, connection=True)
    assert fixed is None


def test_graph_to_smiles_every_outcome(chem, caplog):
    from llamole_amd import molecule_utils as mu
    mols = [
        _graph(["C", "C", "O"], [(0, 1, 1), (1, 2, 2)]),                                  # 0: valid as generated
        _graph(["C", "C", "O", "C"], [(0, 1, 2), (0, 2, 2), (0, 3, 1)]),                  # 1: over-valent carbon, repaired
        _graph(["C", "C", "O"], [(0, 1, 1)]),                                             # 2: two fragments, connected
        _graph(["*", "C", "C", "*"], [(0, 1, 1), (1, 2, 1), (2, 3, 1)]),                  # 3: polymer with legal end points
        _graph(["F", "F", "Cl", "Cl", "Cl"], [(0, 1, 1), (2, 3, 1)]),                     # 4: cannot be connected -> largest fragment
        _graph(["*", "F", "F"], [(0, 1, 1), (1, 2, 1)]),                                  # 5: F with valence 2 even after the repair loop...
    ]
    out = mu.graph_to_smiles(mols, DEC)
    assert len(out) == 6 and all(o is None or isinstance(o, str) for o in out)
    assert out[0] == "C;C;O|0-1:1,1-2:2"
    assert out[1] is not None and chem.MolFromSmiles(out[1]).GetNumAtoms() == 4
    assert out[2] is not None and "." not in out[2] and chem.MolFromSmiles(out[2]).GetNumAtoms() == 3
    assert out[3] is not None and out[3].count("*") == 2
    assert out[4] is not None and "." not in out[4]                                       # the largest fragment only
    assert out[5] is not None                                                             # ... the bond was removed: fragments, largest kept
    # polymerisation points that do not survive capping with H are rejected
    assert mu.check_polymer("*;*|0-1:3") is False and mu.check_polymer("C;C|0-1:1") is True and mu.check_valid("C;O|0-1:2") is True
    assert mu.check_valid("") is False and mu.check_valid("not a molecule") is False and mu.get_mol("Xx") is None
    # an exception inside the pipeline falls back to the unrepaired molecule's SMILES, then to None
    bad = (torch.tensor([99]), torch.zeros(1, 1, dtype=torch.long))                       # class index outside the decoder
    assert mu.graph_to_smiles([bad], DEC) == [None]


def test_graph_to_smiles_needs_rdkit_without_the_double():
    import sys
    from llamole_amd import molecule_utils as mu
    assert "rdkit" not in sys.modules or getattr(sys.modules["rdkit"], "__file__", None) is None
    if "rdkit" in sys.modules:
        pytest.skip("a real or fake rdkit is installed in this process")
    with pytest.raises(ImportError, match="rdkit"):
        mu.graph_to_smiles([_graph(["C"], [])], DEC)


def test_morgan_fingerprint_and_default_template_runner(chem):
    from llamole_amd.graph_predictor import GraphPredictor, _default_template_runner, merge_template_outcomes
    fp = GraphPredictor.smiles_to_fp("C;C;O|0-1:1,1-2:2")
    assert fp.dtype == bool and fp.shape == (2048,) and 0 < fp.sum() < 64
    assert np.array_equal(fp, GraphPredictor.smiles_to_fp("C;C;O|0-1:1,1-2:2"))
    with pytest.raises(ValueError, match="Invalid SMILES"):
        GraphPredictor.smiles_to_fp("not a molecule")
    run = _default_template_runner()                    # rdchiral.main.rdchiralRunText of the double
    reactants, scores, templates = merge_template_outcomes([0.5, 0.3, 0.1, 0.1], ["T_two", "T_none", "T_raise", "T_two"], "P", run)
    assert reactants == ["P_b.a_P"] and templates == ["T_two"] and scores == pytest.approx([1.0])     # the two-reactant outcome, parts sorted


def test_smiles_to_graph_with_the_double(chem):
    import types
    from llamole_amd.modeling_llamole import GraphLLMForCausalMLM
    stub = types.SimpleNamespace()
    g = GraphLLMForCausalMLM.smiles_to_graph(stub, "C;N;[H];*;O|0-1:2,1-2:1,1-3:1,0-4:12")
    assert g.x.tolist() == [4, 5, 117, 6]                          # atomic number - 2, '*' -> 117, hydrogens dropped
    assert g.edge_index.shape == (2, 6) and sorted(g.edge_attr.tolist()) == [1, 1, 2, 2, 4, 4]      # H-N bond dropped, both directions listed
    assert GraphLLMForCausalMLM.smiles_to_graph(stub, "???") is None
    lone = GraphLLMForCausalMLM.smiles_to_graph(stub, "C")
    assert lone.x.tolist() == [4] and lone.edge_index.shape == (2, 0)


def test_fragment_join_copies_bonds_and_skips_unsanitary_joins(chem, monkeypatch):
    from llamole_amd import molecule_utils as mu
    mol = mu.build_molecule(*_graph(["O", "C", "C"], [(1, 2, 2)]), DEC)                   # O + C=C: the second fragment brings its own bond
    joined = mu.connect_fragments(mol)
    assert joined is not None and sorted(int(b.GetBondType()) for b in joined.GetBonds()) == [1, 2]
    calls = {"n": 0}
    real = chem.SanitizeMol

    def first_join_fails(m, *a, **k):
        calls["n"] += 1
        if calls["n"] == 1:
            raise chem.MolSanitizeException("rejected by the test")
        return real(m, *a, **k)
    two = mu.build_molecule(*_graph(["C", "C", "O"], [(0, 1, 1)]), DEC)
    monkeypatch.setattr(chem, "SanitizeMol", first_join_fails)
    joined = mu.connect_fragments(two)
    assert joined is not None and calls["n"] == 2                                         # the next (atom, atom) pair was tried
    assert mu.get_mol(joined) is joined                                                   # a molecule passes through get_mol unchanged


def test_correct_mol_returns_none_when_the_error_cannot_be_parsed(chem, monkeypatch):
    from llamole_amd import molecule_utils as mu
    mol = mu.build_molecule(*_graph(["C", "C"], [(0, 1, 1)]), DEC)
    monkeypatch.setattr(chem, "SanitizeMol", lambda *a, **k: (_ for _ in ()).throw(RuntimeError("not a valence message")))
    assert mu.valency_problem(mol) == (False, [])
    assert mu.correct_mol(mol) == (None, False)


def test_graph_to_smiles_fallbacks(chem, monkeypatch):
    from llamole_amd import molecule_utils as mu
    aromatic_c = _graph(["C", "C", "C", "C"], [(0, 1, 4), (0, 2, 4), (0, 3, 4)])   # unrepairable in both connection modes (only aromatic bonds
    assert mu.graph_to_smiles([aromatic_c], DEC) == [None]            # to lower): the unrepaired molecule is used, and its SMILES does not parse back
    empty = (torch.zeros(0, dtype=torch.long), torch.zeros(0, 0, dtype=torch.long))
    assert mu.graph_to_smiles([empty], DEC) == [None]                 # empty SMILES
    assert mu.graph_to_smiles([_graph(["C"], [])], DEC) == ["C"]      # a one-character largest fragment: the whole SMILES is kept
    monkeypatch.setattr(mu, "correct_mol", lambda *a, **k: (_ for _ in ()).throw(RuntimeError("boom")))
    assert mu.graph_to_smiles([_graph(["C", "O"], [(0, 1, 1)])], DEC) == ["C;O|0-1:1"]      # exception -> SMILES of the unrepaired molecule
    monkeypatch.setattr(chem, "MolToSmiles", lambda m: (_ for _ in ()).throw(RuntimeError("boom too")))
    assert mu.graph_to_smiles([_graph(["C", "O"], [(0, 1, 1)])], DEC) == [None]             # ... and None when that fails as well
    monkeypatch.setattr(mu, "build_molecule", lambda *a, **k: (_ for _ in ()).throw(ImportError("rdkit went away")))
    with pytest.raises(ImportError):
        mu.graph_to_smiles([_graph(["C"], [])], DEC)
