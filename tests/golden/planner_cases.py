"""Scripted expansion tables used to trace the A* planner (reference planner/molstar.py).

Each case is pure data: a target, a set of purchasable molecules, a per-molecule table of
one-step expansions (reactant strings, scores, templates) and per-molecule value estimates.
``make_fns`` turns a case into deterministic ``expand_fn`` / ``value_fn`` callbacks with the
signatures the planner calls them with (expand_fn(mol) -> dict | None; value_fn(mol, parent)).
"""

CASES = {
    # linear two-step route, second candidate of the first expansion is a dead end
    "linear": {
        "target": "T",
        "starting": ["A", "B", "C"],
        "expand": {
            "T": {"reactants": ["I.A", "X"], "scores": [0.7, 0.3], "templates": ["t1", "t2"]},
            "I": {"reactants": ["B.C"], "scores": [0.9], "templates": ["t3"]},
            "X": None,
        },
        "value": {"T": 3.0, "I": 1.5, "X": 0.4, "A": 0.0, "B": 0.0, "C": 0.0},
    },
    # cheaper-looking branch fails; search must back up and take the other one
    "backtrack": {
        "target": "T",
        "starting": ["A", "B"],
        "expand": {
            "T": {"reactants": ["P", "Q.A"], "scores": [0.6, 0.4], "templates": ["t1", "t2"]},
            "P": {"reactants": [], "scores": [], "templates": []},
            "Q": {"reactants": ["A.B", "T.A"], "scores": [0.5, 0.5], "templates": ["t4", "t5"]},
        },
        "value": {"T": 2.0, "P": 0.1, "Q": 1.0, "A": 0.0, "B": 0.0},
    },
    # nothing purchasable is ever reached within the iteration budget
    "fail": {
        "target": "T",
        "starting": ["Z"],
        "iterations": 6,
        "expand": {
            "T": {"reactants": ["U.V"], "scores": [1.0], "templates": ["t1"]},
            "U": {"reactants": ["W"], "scores": [0.2], "templates": ["t2"]},
            "V": None,
            "W": None,
        },
        "value": {"T": 1.0, "U": 0.5, "V": 0.5, "W": 0.3},
    },
    # duplicated reactants inside one outcome, tiny scores (clip at 1e-3), deeper tree
    "deep": {
        "target": "T",
        "starting": ["A", "B", "C", "D"],
        "expand": {
            "T": {"reactants": ["M.M.A", "N"], "scores": [0.0001, 0.9], "templates": ["t1", "t2"]},
            "N": {"reactants": ["O.B"], "scores": [0.8], "templates": ["t3"]},
            "O": {"reactants": ["R.C", "S"], "scores": [0.55, 0.45], "templates": ["t4", "t5"]},
            "R": {"reactants": ["D.A"], "scores": [0.99], "templates": ["t6"]},
            "S": None,
            "M": {"reactants": ["A.B"], "scores": [0.5], "templates": ["t7"]},
        },
        "value": {"T": 5.0, "M": 0.2, "N": 2.0, "O": 1.8, "R": 0.9, "S": 0.1,
                  "A": 0.0, "B": 0.0, "C": 0.0, "D": 0.0},
    },
    # target already purchasable: the planner still searches for a route (mol_tree.py:22-23)
    "known_target": {
        "target": "A",
        "starting": ["A", "B"],
        "expand": {"A": {"reactants": ["B"], "scores": [1.0], "templates": ["t1"]}},
        "value": {"A": 0.0, "B": 0.0},
    },
}


def make_fns(case):
    log = {"expand": [], "value": []}

    def expand_fn(mol):
        log["expand"].append(mol)
        e = case["expand"].get(mol)
        if e is None:
            return None
        return {"reactants": list(e["reactants"]), "scores": list(e["scores"]),
                "templates": list(e["templates"]), "analysis": [len(log["expand"])]}

    def value_fn(mol, parent=None):
        log["value"].append(mol)
        return float(case["value"].get(mol, 1.0))

    return expand_fn, value_fn, log


def _random_case(seed: int, n_targets_unused: int = 0):
    """A seeded random chemistry as a pure table: hash-derived outcomes per molecule (1-4 outcomes of 1-2 reactants: purchasable,
    the product itself -> cycle, or an intermediate out of 40), hash-derived scores / values, dead ends; closed under expansion."""
    import zlib

    def h(*a):
        return zlib.crc32(("|".join(map(str, a)) + f"#{seed}").encode())

    starting = [f"S{i}" for i in range(12)]
    target = f"M{40 + seed}"
    expand, value, todo = {}, {}, [target]
    while todo:
        mol = todo.pop()
        if mol in expand or mol in starting:
            continue
        value[mol] = (h("v", mol) % 7000) / 1000.0 + 0.0001 * (h("w", mol) % 997)      # distinct: no argmin ties
        k = h("n", mol) % 5
        if k == 0:
            expand[mol] = None
            continue
        reactants, scores, templates = [], [], []
        for j in range(k):
            parts = []
            for r in range(1 + h("a", mol, j) % 2):
                u = h("r", mol, j, r)
                parts.append(f"S{u % 12}" if u % 3 == 0 else (mol if u % 17 == 0 else f"M{u % 40}"))
            reactants.append(".".join(parts))
            scores.append(((h("s", mol, j) % 1000) + 1) / 1000.0)
            templates.append(f"T{h('t', mol, j) % 30}")
            todo.extend(p for p in parts if p not in expand)
        expand[mol] = {"reactants": reactants, "scores": scores, "templates": templates}
    for s in starting:
        value[s] = 0.0
    return {"target": target, "starting": starting, "iterations": 8, "expand": expand, "value": value}


# The reference de-duplicates the reactants of an outcome through a set (planner/molstar.py:54): whenever two open nodes tie, the order in
# which it creates them -- Python's string hash order -- decides its trace ("fail" above ties U and V; its committed trace is the one of
# PYTHONHASHSEED=1, equal to insertion order).  Of 24 seeded random chemistries these are the non-trivial ones whose reference trace was
# identical under PYTHONHASHSEED = 0..11 (1, 7, 11, 12 were not and are left out):
_STABLE_SEEDS = (2, 3, 4, 5, 10, 15, 16, 20, 22)
CASES.update({f"random_{seed}": _random_case(seed) for seed in _STABLE_SEEDS})
