"""A small test double for the parts of ``rdkit.Chem`` / ``rdkit.Chem.AllChem`` / ``rdkit.RDLogger`` / ``rdchiral.main`` that the host
chemistry tails call (llamole_amd/molecule_utils.py, graph_predictor.py, modeling_llamole.smiles_to_graph).

rdkit and rdchiral are in neither image, so those tails had never EXECUTED anywhere.  This double lets every line
run: it models atoms, bonds, formal charges, explicit / implicit hydrogens, a valence table with RDKit's sanitisation error text
("Explicit valence for atom # 3 N, 4, is greater than permitted"), connected components, and a linear molecule notation that
``MolToSmiles`` writes and ``MolFromSmiles`` reads back.  It PINS NOTHING: it is not RDKit, its "SMILES" are not SMILES, and agreement
with it says only that the control flow of the restated code is sound.  a15 / f3 stay "partial" until the real libraries are available.
"""
from __future__ import annotations

import sys
import types
import zlib

SYMBOL_Z = {"*": 0, "H": 1, "B": 5, "C": 6, "N": 7, "O": 8, "F": 9, "Na": 11, "Si": 14, "P": 15, "S": 16, "Cl": 17, "Ge": 32, "Se": 34, "Br": 35,
            "Sn": 50, "I": 53}
ALLOWED = {0: [99], 1: [1], 5: [3], 6: [4], 7: [3], 8: [2], 9: [1], 11: [1], 14: [4], 15: [3, 5], 16: [2, 4, 6], 17: [1], 32: [4], 34: [2, 4, 6],
           35: [1], 50: [2, 4], 53: [1]}


class _BondType:
    def __init__(self, name, value, order):
        self.name, self.value, self.order = name, value, order

    def __int__(self):
        return self.value

    def __str__(self):
        return self.name

    def __repr__(self):
        return f"rdkit.Chem.rdchem.BondType.{self.name}"


class BondType:
    SINGLE = _BondType("SINGLE", 1, 1.0)
    DOUBLE = _BondType("DOUBLE", 2, 2.0)
    TRIPLE = _BondType("TRIPLE", 3, 3.0)
    AROMATIC = _BondType("AROMATIC", 12, 1.5)
    BY_VALUE = {1: SINGLE, 2: DOUBLE, 3: TRIPLE, 12: AROMATIC}


class MolSanitizeException(ValueError):
    pass


class AtomValenceException(MolSanitizeException):
    pass


class Atom:
    def __init__(self, symbol):
        cache = None
        if isinstance(symbol, Atom):
            symbol, charge, hs, cache = symbol.symbol, symbol.charge, symbol.explicit_hs, symbol._implicit_cache
        else:
            charge, hs = 0, 0
        if symbol not in SYMBOL_Z:
            raise ValueError(f"unknown element {symbol!r}")
        self.symbol, self.charge, self.explicit_hs = symbol, charge, hs
        # RDKit keeps the implicit-H count in a property cache that is filled by sanitisation and TRAVELS with a copied atom; the
        # reference's fragment join relies on it (the H count it reads after adding the joining bond is the fragment's, not the trial's)
        self._implicit_cache = cache
        self.mol, self.idx = None, -1

    def GetAtomicNum(self):
        return SYMBOL_Z[self.symbol]

    def GetSymbol(self):
        return self.symbol

    def GetIdx(self):
        return self.idx

    def GetFormalCharge(self):
        return self.charge

    def SetFormalCharge(self, c):
        self.charge = int(c)

    def SetNumExplicitHs(self, n):
        self.explicit_hs = int(n)

    def GetBonds(self):
        return [b for b in self.mol.bonds if self.idx in (b.a, b.b)]

    def _explicit_valence(self):
        return sum(b.type.order for b in self.GetBonds()) + self.explicit_hs

    def _max_valence(self):
        z = self.GetAtomicNum()
        extra = self.charge if z in (7, 8, 15, 16) else (-self.charge if z == 5 else 0)
        return [v + extra for v in ALLOWED[z]]

    def _compute_implicit(self):
        ev = self._explicit_valence()
        for v in self._max_valence():
            if v >= ev:
                return int(v - ev) if self.GetAtomicNum() != 0 else 0
        return 0

    def GetImplicitValence(self):
        if self._implicit_cache is None:
            self._implicit_cache = self._compute_implicit()
        return self._implicit_cache

    def GetTotalNumHs(self):
        return self.explicit_hs + self.GetImplicitValence()


class Bond:
    def __init__(self, mol, a, b, btype):
        self.mol, self.a, self.b, self.type = mol, a, b, btype

    def GetIdx(self):
        return self.mol.bonds.index(self)

    def GetBondType(self):
        return self.type

    def GetBeginAtomIdx(self):
        return self.a

    def GetEndAtomIdx(self):
        return self.b


class Mol:
    def __init__(self, other=None):
        self.atoms, self.bonds = [], []
        if other is not None:
            for a in other.atoms:
                self.AddAtom(a)
            for b in other.bonds:
                self.bonds.append(Bond(self, b.a, b.b, b.type))

    def AddAtom(self, atom):
        a = Atom(atom)
        a.mol, a.idx = self, len(self.atoms)
        self.atoms.append(a)
        return a.idx

    def AddBond(self, i, j, btype):
        if i == j or any({b.a, b.b} == {i, j} for b in self.bonds):
            raise RuntimeError("bond already exists or is a self loop")
        self.bonds.append(Bond(self, int(i), int(j), btype))
        return len(self.bonds)

    def RemoveBond(self, i, j):
        self.bonds = [b for b in self.bonds if {b.a, b.b} != {i, j}]

    def GetAtomWithIdx(self, i):
        return self.atoms[int(i)]

    def GetAtoms(self):
        return list(self.atoms)

    def GetBonds(self):
        return list(self.bonds)

    def GetNumAtoms(self):
        return len(self.atoms)


RWMol = Mol


class SanitizeFlags:
    SANITIZE_PROPERTIES = 2
    SANITIZE_ALL = 0xFFFF


def SanitizeMol(mol, sanitizeOps=SanitizeFlags.SANITIZE_ALL):
    if mol is None:
        raise ValueError("None molecule")
    for a in mol.atoms:
        ev = a._explicit_valence()
        if ev > max(a._max_valence()) + 1e-9:
            raise AtomValenceException(f"Explicit valence for atom # {a.idx} {a.symbol}, {int(round(ev))}, is greater than permitted")
    for a in mol.atoms:
        a._implicit_cache = a._compute_implicit()      # updatePropertyCache
    return 0


def GetMolFrags(mol, asMols=False, sanitizeFrags=True):
    n = len(mol.atoms)
    comp = list(range(n))

    def find(i):
        while comp[i] != i:
            comp[i] = comp[comp[i]]
            i = comp[i]
        return i
    for b in mol.bonds:
        comp[find(b.a)] = find(b.b)
    groups = {}
    for i in range(n):
        groups.setdefault(find(i), []).append(i)
    frags = sorted(groups.values(), key=lambda g: g[0])
    if not asMols:
        return tuple(tuple(g) for g in frags)
    out = []
    for g in frags:
        m, remap = Mol(), {}
        for i in g:
            remap[i] = m.AddAtom(mol.atoms[i])
        for b in mol.bonds:
            if b.a in remap:
                m.AddBond(remap[b.a], remap[b.b], b.type)
        out.append(m)
    return tuple(out)


def _atom_token(a):
    t = a.symbol
    if a.charge:
        t = f"[{t}{'+' if a.charge > 0 else '-'}{abs(a.charge) if abs(a.charge) > 1 else ''}]"
    if a.explicit_hs:
        t += f"h{a.explicit_hs}"
    return t


def MolToSmiles(mol):
    """One fragment = `atom;atom;...|i-j:t,i-j:t` (atom order kept), fragments joined by '.'."""
    if mol is None:
        raise ValueError("None molecule")
    parts = []
    for frag in GetMolFrags(mol, asMols=True, sanitizeFrags=False):
        atoms = ";".join(_atom_token(a) for a in frag.atoms)
        bonds = ",".join(f"{b.a}-{b.b}:{b.type.value}" for b in sorted(frag.bonds, key=lambda b: (min(b.a, b.b), max(b.a, b.b))))
        parts.append(atoms + ("|" + bonds if bonds else ""))
    return ".".join(parts)


def MolFromSmiles(s):
    try:
        mol = Mol()
        for part in s.split("."):
            atoms, _, bonds = part.partition("|")
            base = len(mol.atoms)
            for tok in atoms.split(";"):
                hs = 0
                if "h" in tok and not tok.startswith("[H"):
                    tok, _, h = tok.rpartition("h")
                    hs = int(h)
                charge = 0
                if tok.startswith("["):
                    body = tok[1:-1]
                    sym = body.rstrip("+-0123456789")
                    tail = body[len(sym):]
                    if tail:
                        charge = (1 if tail[0] == "+" else -1) * (int(tail[1:]) if len(tail) > 1 else 1)
                else:
                    sym = tok
                a = Atom(sym)
                a.charge, a.explicit_hs = charge, hs
                mol.AddAtom(a)
            for bt in filter(None, bonds.split(",")):
                ij, _, t = bt.partition(":")
                i, _, j = ij.partition("-")
                mol.AddBond(base + int(i), base + int(j), BondType.BY_VALUE[int(t)])
        return mol if mol.atoms else None
    except Exception:       # noqa: BLE001 -- RDKit returns None for unparsable input
        return None


class _Fingerprint:
    def __init__(self, bits, n):
        self.bits, self.n = sorted(bits), n

    def GetNumBits(self):
        return self.n

    def GetOnBits(self):
        return list(self.bits)


def GetMorganFingerprintAsBitVect(mol, radius, nBits=2048):
    bits = set()
    for a in mol.atoms:
        env = a.symbol
        for r in range(radius + 1):
            bits.add(zlib.crc32(f"{env}|{r}|{a._explicit_valence()}".encode()) % nBits)
            env += "".join(sorted(mol.atoms[b.b if b.a == a.idx else b.a].symbol for b in a.GetBonds()))
    return _Fingerprint(bits, nBits)


def split_template_runner(template: str, smiles: str):
    """A scripted `rdchiralRunText` whose outcomes are molecules of this double: the product with its last bond cut (two reactants when
    that disconnects it); every seventh template does not apply, every eleventh raises."""
    i = int("".join(ch for ch in template if ch.isdigit()) or 0)
    if i % 7 == 0:
        return []
    if i % 11 == 0:
        raise RuntimeError("template does not apply")
    mol = MolFromSmiles(smiles)
    if mol is None or not mol.bonds:
        return []
    cut = mol.bonds[-1 - (i % len(mol.bonds))]
    mol.RemoveBond(cut.a, cut.b)
    return [MolToSmiles(mol)]


class _Setter:
    """monkeypatch.setitem stand-in for processes that install the double for their whole lifetime (rank workers of the eval test)."""

    @staticmethod
    def setitem(mapping, key, value):
        mapping[key] = value


def install_global(template_outcomes=None):
    return install(_Setter, template_outcomes)


def install(monkeypatch, template_outcomes=None):
    """Put the double into sys.modules as rdkit / rdkit.Chem / rdkit.Chem.AllChem / rdkit.RDLogger (+ rdchiral.main when
    `template_outcomes` -- a callable (template, smiles) -> list of reactant strings -- is given)."""
    chem = types.ModuleType("rdkit.Chem")
    for name in ("Atom", "Mol", "RWMol", "BondType", "SanitizeFlags", "SanitizeMol", "GetMolFrags", "MolToSmiles", "MolFromSmiles",
                 "MolSanitizeException", "AtomValenceException"):
        setattr(chem, name, globals()[name])
    chem.rdchem = types.SimpleNamespace(BondType=BondType, Atom=Atom, Mol=Mol, RWMol=RWMol)
    chem.rdmolops = types.SimpleNamespace(GetMolFrags=GetMolFrags)
    allchem = types.ModuleType("rdkit.Chem.AllChem")
    allchem.GetMorganFingerprintAsBitVect = GetMorganFingerprintAsBitVect
    chem.AllChem = allchem
    rdlogger = types.ModuleType("rdkit.RDLogger")
    rdlogger.DisableLog = lambda *_a, **_k: None
    rdkit = types.ModuleType("rdkit")
    rdkit.Chem, rdkit.RDLogger = chem, rdlogger
    for name, mod in (("rdkit", rdkit), ("rdkit.Chem", chem), ("rdkit.Chem.AllChem", allchem), ("rdkit.RDLogger", rdlogger)):
        monkeypatch.setitem(sys.modules, name, mod)
    if template_outcomes is not None:
        rdchiral = types.ModuleType("rdchiral")
        main = types.ModuleType("rdchiral.main")
        main.rdchiralRunText = template_outcomes
        rdchiral.main = main
        monkeypatch.setitem(sys.modules, "rdchiral", rdchiral)
        monkeypatch.setitem(sys.modules, "rdchiral.main", main)
    return chem
