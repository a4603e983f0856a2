"""Host chemistry tail of GraphDiT.generate: integer graph -> repaired RDKit molecule -> SMILES.

Restates reference ``src/model/graph_decoder/molecule_utils.py:49-352`` (third-party dependency
``rdkit==2023.9.6``, reference requirements.txt:22).  This is CPU work per molecule and stays host code; the
parity / throughput unit of the MI355X path is the integer graph ``(atom_types[n], edge_types[n,n])`` that
feeds this module.  rdkit is imported lazily: without it ``graph_to_smiles`` / ``check_valid`` raise
ImportError (never a silent fallback).  rdkit is absent from the build container, so this module is
UNPINNED against the reference; the steps and constants follow the reference lines cited.
"""
from __future__ import annotations

import logging
import re
from typing import List, Optional, Sequence, Tuple

logger = logging.getLogger(__name__)

# default valences used by the +1 formal-charge repair (molecule_utils.py:35)
ATOM_VALENCY = {6: 4, 7: 3, 8: 2, 9: 1, 15: 3, 16: 2, 17: 1, 35: 1, 53: 1}


def _chem():
    try:
        from rdkit import Chem, RDLogger
    except ImportError as e:  # pragma: no cover - depends on the environment
        raise ImportError("graph -> SMILES needs `rdkit` (reference requirements.txt:22); "
                          "use GraphDiT.generate_graphs() for the integer graphs") from e
    RDLogger.DisableLog("rdApp.*")
    return Chem


def _bond_types(Chem):
    bt = Chem.rdchem.BondType
    return [None, bt.SINGLE, bt.DOUBLE, bt.TRIPLE, bt.AROMATIC]     # edge class -> bond (molecule_utils.py:27-33)


def get_mol(smiles_or_mol):
    Chem = _chem()
    if isinstance(smiles_or_mol, str):
        if not smiles_or_mol:
            return None
        mol = Chem.MolFromSmiles(smiles_or_mol)
        if mol is None:
            return None
        try:
            Chem.SanitizeMol(mol)
        except ValueError:
            return None
        return mol
    return smiles_or_mol


def mol2smiles(mol) -> Optional[str]:
    Chem = _chem()
    if mol is None:
        return None
    try:
        Chem.SanitizeMol(mol)
    except ValueError:
        return None
    return Chem.MolToSmiles(mol)


def check_valid(smiles) -> bool:
    """molecule_utils.py:213-220."""
    mol = get_mol(smiles)
    return mol is not None and mol2smiles(mol) is not None


def check_polymer(smiles: str) -> bool:
    """Polymerisation points '*' must still give a valid monomer when capped with H (molecule_utils.py:39-47)."""
    if "*" in smiles:
        return mol2smiles(get_mol(smiles.replace("*", "[H]"))) is not None
    return True


def valency_problem(mol) -> Tuple[bool, Optional[List[int]]]:
    """(ok, [atom_idx, valence]) parsed from RDKit's sanitisation error (molecule_utils.py:246-259)."""
    Chem = _chem()
    try:
        Chem.SanitizeMol(mol, sanitizeOps=Chem.SanitizeFlags.SANITIZE_PROPERTIES)
        return True, None
    except ValueError as e:
        msg = str(e)
        return False, [int(t) for t in re.findall(r"\d+", msg[msg.find("#"):])]
    except Exception:
        return False, []


def build_molecule(atom_types, edge_types, atom_decoder: Sequence[str]):
    """RWMol from the integer graph; N/O/S atoms one over their valence get a +1 formal charge
    (molecule_utils.py:113-166)."""
    Chem = _chem()
    bonds = _bond_types(Chem)
    mol = Chem.RWMol()
    for a in atom_types:
        mol.AddAtom(Chem.Atom(atom_decoder[int(a)]))
    n = len(atom_types)
    for i in range(n):
        for j in range(i + 1, n):
            k = int(edge_types[i][j])
            if k <= 0:
                continue
            mol.AddBond(i, j, bonds[k])
            ok, info = valency_problem(mol)
            if ok or info is None or len(info) != 2:
                continue
            idx, v = info
            an = mol.GetAtomWithIdx(idx).GetAtomicNum()
            if an in (7, 8, 16) and v - ATOM_VALENCY[an] == 1:
                mol.GetAtomWithIdx(idx).SetFormalCharge(1)
    return mol


def _free_valence_atoms(frag):
    return [a for a in frag.GetAtoms() if a.GetAtomicNum() > 1 and a.GetImplicitValence() > 0]


def _try_join(Chem, base, frag, a1, a2):
    trial = Chem.RWMol(base)
    remap = {a.GetIdx(): trial.AddAtom(a) for a in Chem.RWMol(frag).GetAtoms()}
    trial.AddBond(a1.GetIdx(), remap[a2.GetIdx()], Chem.BondType.SINGLE)
    for idx in (a1.GetIdx(), remap[a2.GetIdx()]):
        atom = trial.GetAtomWithIdx(idx)
        atom.SetNumExplicitHs(max(0, atom.GetTotalNumHs() - 1))
    for b in frag.GetBonds():
        trial.AddBond(remap[b.GetBeginAtomIdx()], remap[b.GetEndAtomIdx()], b.GetBondType())
    out = Chem.Mol(trial)
    try:
        Chem.SanitizeMol(out)
        return out
    except Chem.MolSanitizeException:
        return None


def connect_fragments(mol):
    """Join disconnected fragments with single bonds between atoms that still have free valence
    (molecule_utils.py:322-352); None when some fragment cannot be attached."""
    Chem = _chem()
    frags = Chem.GetMolFrags(mol, asMols=True, sanitizeFrags=False)
    if len(frags) < 2:
        return mol
    combined = Chem.RWMol(frags[0])
    for frag in frags[1:]:
        joined = None
        for a1 in _free_valence_atoms(combined):
            for a2 in _free_valence_atoms(frag):
                joined = _try_join(Chem, combined, frag, a1, a2)
                if joined is not None:
                    break
            if joined is not None:
                break
        if joined is None:
            return None
        combined = joined
    return combined


def correct_mol(mol, connection: bool = False):
    """Iteratively lower the order of the highest-order bond at the offending atom until valences are legal
    (molecule_utils.py:169-210).  Returns (mol | None, was_already_valid)."""
    Chem = _chem()
    bonds = _bond_types(Chem)
    already_ok, _ = valency_problem(mol)
    while True:
        if connection:
            mol = connect_fragments(mol)
            if mol is None:
                return None, already_ok
        ok, info = valency_problem(mol)
        if ok:
            return mol, already_ok
        try:
            assert len(info) == 2
            idx = info[0]
            queue, n_arom = [], 0
            for b in mol.GetAtomWithIdx(idx).GetBonds():
                t = int(b.GetBondType())
                queue.append((b.GetIdx(), t, b.GetBeginAtomIdx(), b.GetEndAtomIdx()))
                n_arom += 1 if t == 12 else 0
            queue.sort(key=lambda q: q[1], reverse=True)
            if queue[-1][1] == 12:
                return None, already_ok
            _, t, start, end = queue[n_arom]
            mol.RemoveBond(start, end)
            if t - 1 >= 1:
                mol.AddBond(start, end, bonds[t - 1])
        except Exception:
            return None, already_ok


def graph_to_smiles(molecule_list, atom_decoder) -> List[Optional[str]]:
    """molecule_utils.py:49-111: repair, SMILES, largest fragment, polymer check; invalid -> None."""
    Chem = _chem()
    out: List[Optional[str]] = []
    for index, (atom_types, edge_types) in enumerate(molecule_list):
        mol_init = None
        try:
            mol_init = build_molecule(atom_types, edge_types, atom_decoder)
            mol_fixed = None
            for connection in (True, False):
                mol_fixed, _ = correct_mol(mol_init, connection=connection)
                if mol_fixed is not None:
                    break
            if mol_fixed is None:
                mol_fixed = mol_init
            smiles = mol2smiles(mol_fixed) or Chem.MolToSmiles(mol_fixed)
            if not smiles:
                out.append(None)
                continue
            mol = get_mol(smiles)
            if mol is None:
                out.append(None)
                continue
            frags = Chem.rdmolops.GetMolFrags(mol, asMols=True, sanitizeFrags=False)
            largest = mol2smiles(max(frags, key=lambda m: m.GetNumAtoms()))
            if largest and len(largest) > 1:
                out.append(largest if check_polymer(largest) else None)
            else:
                out.append(smiles if check_polymer(smiles) else None)
        except ImportError:
            raise
        except Exception as e:
            logger.error("Error processing molecule %d: %s", index, e)
            try:
                fb = Chem.MolToSmiles(mol_init) if mol_init is not None else None
                out.append(fb if fb else None)
            except Exception:
                out.append(None)
    return out
