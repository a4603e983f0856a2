"""Retro*-style A* search over an AND-OR tree (host control loop of retrosynthesis).

Restates reference ``src/model/planner/{molstar,mol_tree,mol_node,reaction_node,syn_route}.py`` in one
module.  Same call contract -- ``molstar(target_mol, target_mol_id, starting_mols, expand_fn, value_fn,
iterations, max_time) -> (success, route, n_iterations)`` with ``expand_fn(mol) -> dict | None`` and
``value_fn(mol, parent_reaction) -> float`` -- and the same value algebra:

  OR node (molecule)   V(m) = min over child reactions, 0 if purchasable;  open until expanded
  AND node (reaction)  V(r) = cost(r) + sum V(reactants);  V_target(r) = V_target(parent) - V(parent) + V(r)
  pick the open molecule with the smallest V_target, expand, push value deltas up and sideways.

This is microseconds of Python per iteration; wall time is spent in the callbacks (LLM decode + GIN
predictor), so it stays host code.  Two deliberate differences from the reference, neither changes a
result: purchasable molecules are looked up in a ``set`` (the reference scans a list, mol_tree.py:27), and
duplicate reactants inside one outcome are removed keeping first-seen order (the reference uses
``list(set(...))``, whose order depends on string hashing, molstar.py:54).
"""
from __future__ import annotations

import math
import time
from collections import deque
from typing import Callable, Dict, Iterable, List, Optional, Sequence, Tuple

INF = math.inf


class Molecule:
    """OR node."""

    __slots__ = ("mol", "value", "pred_value", "succ_value", "parent", "depth", "known", "children", "succ",
                 "open", "id")

    def __init__(self, mol: str, estimate: float, parent: Optional["Reaction"], known: bool):
        self.mol = mol
        self.pred_value = estimate
        self.value = estimate
        self.succ_value = INF
        self.parent = parent
        self.depth = 0 if parent is None else parent.depth
        self.known = known
        self.children: List["Reaction"] = []
        self.succ = known
        self.open = not known
        self.id = -1
        if known:               # purchasable: nothing left to pay
            self.value = 0
            self.succ_value = 0
        if parent is not None:
            parent.children.append(self)

    def target_value(self) -> float:
        """V_target(m | tree): cost of the cheapest full plan that goes through this molecule."""
        return self.value if self.parent is None else self.parent.target

    def ancestors(self) -> set:
        out = {self.mol}
        node = self
        while node.parent is not None:
            node = node.parent.parent
            out.add(node.mol)
        return out

    def close(self, allow_childless: bool = False) -> float:
        """First evaluation after expansion; returns the change of V(m)."""
        assert self.open and (allow_childless or self.children)
        best = INF
        self.succ = False
        for r in self.children:
            best = min(best, r.value)
            self.succ = self.succ or r.succ
        delta = best - self.value
        self.value = best
        if self.succ:
            for r in self.children:
                self.succ_value = min(self.succ_value, r.succ_value)
        self.open = False
        return delta

    def refresh(self, child_succ: bool):
        """A child reaction changed: recompute and, if anything moved, keep pushing upwards."""
        assert not self.known
        best = INF
        for r in self.children:
            best = min(best, r.value)
        succ = self.succ or child_succ
        changed = (best != self.value) or (succ != self.succ)
        succ_value = INF
        if succ:
            for r in self.children:
                succ_value = min(succ_value, r.succ_value)
            changed = changed or (succ_value != self.succ_value)
        delta = best - self.value
        self.value, self.succ, self.succ_value = best, succ, succ_value
        if changed and self.parent is not None:
            self.parent.absorb(delta, came_from=self.mol)


class Reaction:
    """AND node."""

    __slots__ = ("parent", "depth", "cost", "template", "analysis_tokens", "children", "value", "succ_value",
                 "target", "succ", "open", "id")

    def __init__(self, parent: Molecule, cost: float, template, analysis_tokens):
        self.parent = parent
        self.depth = parent.depth + 1
        self.cost = cost
        self.template = template
        self.analysis_tokens = analysis_tokens
        self.children: List[Molecule] = []
        self.value = None
        self.succ_value = INF
        self.target = None
        self.succ = None
        self.open = True
        self.id = -1
        parent.children.append(self)

    def _resolve_success(self):
        self.succ = all(m.succ for m in self.children)
        if self.succ:
            self.succ_value = self.cost + sum(m.succ_value for m in self.children)

    def close(self):
        assert self.open
        self.value = self.cost + sum(m.value for m in self.children)
        self._resolve_success()
        self.target = self.parent.target_value() - self.parent.value + self.value
        self.open = False

    def absorb(self, delta: float, came_from: Optional[str] = None):
        self.value += delta
        self.target += delta
        self._resolve_success()
        if delta != 0:
            assert came_from
            self.spread(delta, skip=came_from)
        self.parent.refresh(self.succ)

    def spread(self, delta: float, skip: Optional[str] = None):
        """Sibling sub-trees see the same change in their V_target."""
        if skip is None:
            self.target += delta
        for m in self.children:
            if skip is None or m.mol != skip:
                for r in m.children:
                    r.spread(delta)


class Route:
    """Best successful plan, flattened breadth-first (reference syn_route.py)."""

    def __init__(self, target: str, succ_value: float, search_status: float):
        self.target_mol = target
        self.mols = [target]
        self.values: List[Optional[float]] = [None]
        self.templates: List[Optional[str]] = [None]
        self.parents = [-1]
        self.children: List[Optional[List[int]]] = [None]
        self.costs: Dict[int, float] = {}
        self.analysis_dict: Dict[int, object] = {}
        self.succ_value = succ_value
        self.search_status = search_status
        self.optimal = succ_value <= search_status
        self.total_cost = 0
        self.length = 0

    def set_value(self, mol: str, value: float):
        self.values[self.mols.index(mol)] = value

    def add_reaction(self, mol, value, template, analysis_tokens, reactants, cost):
        self.total_cost += cost
        self.length += 1
        pid = self.mols.index(mol)
        self.values[pid] = value
        self.templates[pid] = template
        self.children[pid] = []
        self.costs[pid] = cost
        self.analysis_dict[pid] = analysis_tokens
        for r in reactants:
            self.mols.append(r)
            self.values.append(None)
            self.templates.append(None)
            self.parents.append(pid)
            self.children.append(None)
            self.children[pid].append(len(self.mols) - 1)

    def _reaction_at(self, i: int):
        text = self.mols[i] + ">>" + ".".join(self.mols[c] for c in self.children[i])
        return text, math.exp(-self.costs[i]), self.analysis_dict[i], self.templates[i]

    def get_reaction_list(self):
        """(reactions, templates, exp(-cost) per reaction, analysis tokens) in route order."""
        reactions, templates, costs, analyses = [], [], [], []
        for i in range(len(self.mols)):
            if i == 0 or self.children[i] is not None:
                r, c, a, t = self._reaction_at(i)
                reactions.append(r)
                costs.append(c)
                analyses.append(a)
                templates.append(t)
        return reactions, templates, costs, analyses

    def get_template_list(self):
        return self.templates


class _MolView:
    __slots__ = ("mol",)

    def __init__(self, mol: str):
        self.mol = mol


class ReactionView:
    """What ``value_fn`` may read of a parent reaction (``depth``, ``template``, ``children[i].mol``) at the moment the
    reference evaluates a new reactant: the reaction exists and holds only the reactants attached BEFORE this one
    (mol_tree.py:25-33 evaluates, then MolNode.__init__ appends).  Lets a whole expansion be evaluated in one batch."""
    __slots__ = ("depth", "template", "children")

    def __init__(self, depth: int, template, earlier_reactants: Sequence[str]):
        self.depth = depth
        self.template = template
        self.children = [_MolView(m) for m in earlier_reactants]


class SearchTree:
    def __init__(self, target_mol: str, known_mols: Iterable[str], value_fn: Callable,
                 value_batch_fn: Optional[Callable] = None, root_estimate: Optional[float] = None):
        self.target_mol = target_mol
        self.known = known_mols if isinstance(known_mols, (set, frozenset)) else set(known_mols)
        self.value_fn = value_fn
        # optional: value_batch_fn([(mol, ReactionView), ...]) -> [float, ...] evaluates every new, non-purchasable
        # reactant of one expansion in ONE call (the reference pays one LLM forward per node).  The
        # estimate of a purchasable molecule is never read (its value is 0), so those are not requested.
        self.value_batch_fn = value_batch_fn
        self.mol_nodes: List[Molecule] = []
        self.reaction_nodes: List[Reaction] = []
        self.root = self._new_mol(target_mol, None, root_estimate)
        self.succ = False
        self.search_status = 0

    def _new_mol(self, mol: str, parent: Optional[Reaction], estimate: Optional[float] = None) -> Molecule:
        if estimate is None:
            estimate = self.value_fn(mol, parent)   # one LLM forward per new tree node in Llamole
        node = Molecule(mol, estimate, parent, mol in self.known)
        self.mol_nodes.append(node)
        node.id = len(self.mol_nodes)
        return node

    def _dead_end(self, node: Molecule) -> bool:
        assert node.close(allow_childless=True) == INF
        if node.parent is not None:
            node.parent.absorb(INF, came_from=node.mol)
        return self.succ

    def value_requests(self, node: Molecule, reactant_lists, templates) -> list:
        """The ``(mol, ReactionView)`` pairs whose estimates ``expand`` will read, in the order it reads them."""
        lineage = node.ancestors()
        requests = []
        for i in range(len(reactant_lists)):
            if any(m in lineage for m in reactant_lists[i]):
                continue
            for k, m in enumerate(reactant_lists[i]):
                if m not in self.known:
                    requests.append((m, ReactionView(node.depth + 1, templates[i], reactant_lists[i][:k])))
        return requests

    def expand(self, node: Molecule, reactant_lists, costs, templates, analysis_tokens, estimates=None) -> bool:
        """``estimates``: the values of ``value_requests(node, reactant_lists, templates)`` when the caller evaluated them already
        (molstar_many: one evaluation call for all searches of a round)."""
        assert not node.known and not node.children
        if costs is None:
            return self._dead_end(node)
        assert node.open
        lineage = node.ancestors()
        if estimates is not None:
            estimates = iter(estimates)
        elif self.value_batch_fn is not None:
            requests = self.value_requests(node, reactant_lists, templates)
            estimates = iter(self.value_batch_fn(requests)) if requests else iter(())
        for i in range(len(costs)):
            assert costs[i] >= 0
            if any(m in lineage for m in reactant_lists[i]):
                continue                                   # would re-introduce an ancestor: cycle
            rxn = Reaction(node, costs[i], templates[i], analysis_tokens)
            for m in reactant_lists[i]:
                if estimates is None:
                    self._new_mol(m, rxn)
                else:
                    self._new_mol(m, rxn, 0.0 if m in self.known else float(next(estimates)))
            rxn.close()
            self.reaction_nodes.append(rxn)
            rxn.id = len(self.reaction_nodes)
        if not node.children:
            return self._dead_end(node)
        delta = node.close()
        if node.parent is not None:
            node.parent.absorb(delta, came_from=node.mol)
        if not self.succ and self.root.succ:
            self.succ = True
        return self.succ

    def best_route(self) -> Optional[Route]:
        if not self.succ:
            return None
        route = Route(self.root.mol, self.root.succ_value, self.search_status)
        queue = deque([self.root])
        while queue:
            m = queue.popleft()
            if m.known:
                route.set_value(m.mol, m.succ_value)
                continue
            best = None
            for r in m.children:
                if r.succ and (best is None or r.succ_value < best.succ_value):
                    best = r
            assert best.succ_value == m.succ_value
            queue.extend(best.children)
            route.add_reaction(m.mol, m.succ_value, best.template, best.analysis_tokens,
                               [c.mol for c in best.children], best.cost)
        return route


class MolStarSearch:
    """One A* search as an explicit state machine: ``select()`` -> the open molecule to expand (or None when the search
    is over), ``apply(result)`` -> feed the expansion back.  ``molstar`` drives one of these; ``molstar_many`` drives
    several in lock step so that their expansions can share one batched LLM decode / GIN forward."""

    def __init__(self, target_mol, starting_mols, value_fn, iterations, max_time=300, value_batch_fn=None, root_estimate=None):
        self.tree = SearchTree(target_mol, starting_mols, value_fn, value_batch_fn, root_estimate)
        self.iterations = iterations
        self.max_time = max_time
        self.t0 = time.time()
        self.done_iters = 0
        self.finished = self.tree.succ
        self._node = None

    def select(self, now: Optional[float] = None) -> Optional[Molecule]:
        """``now``: the wall-clock reading to judge ``max_time`` by (default: this process's own clock).  ``molstar_many`` reads the
        clock ONCE per round for all its searches, and a driver that replicates the searches on several ranks hands in a clock
        every rank agrees on -- a rank that timed out one round before the others would skip a collective they are waiting in."""
        if self.finished:
            return None
        if self.done_iters >= self.iterations:
            self.finished = True
            return None
        self.done_iters += 1          # iterations ENTERED, like the reference's `for done in range(iterations)` + 1
        if (time.time() if now is None else now) - self.t0 > self.max_time:
            self.finished = True
            return None
        best_node, best_score = None, INF
        for m in self.tree.mol_nodes:                     # first minimum wins, like np.argmin
            if m.open:
                score = m.target_value()
                if score < best_score:
                    best_node, best_score = m, score
        if best_node is None:
            self.finished = True
            return None
        self.tree.search_status = best_score
        self._node = best_node
        return best_node

    @staticmethod
    def _reactant_lists(result):
        return [list(dict.fromkeys(result["reactants"][j].split("."))) for j in range(len(result["scores"]))]

    def value_requests(self, result) -> list:
        """What ``apply(result)`` would ask ``value_batch_fn`` for (empty for a dead end)."""
        if result is None or len(result["scores"]) == 0:
            return []
        return self.tree.value_requests(self._node, self._reactant_lists(result), result["templates"])

    def apply(self, result, estimates=None) -> None:
        node, tree = self._node, self.tree
        self._node = None
        if result is not None and len(result["scores"]) > 0:
            scores = result["scores"]
            costs = [-math.log(min(max(float(s), 1e-3), 1.0)) for s in scores]
            reactant_lists = self._reactant_lists(result)
            if tree.expand(node, reactant_lists, costs, result["templates"], result["analysis"], estimates):
                self.finished = True
            elif tree.root.succ_value <= tree.search_status:
                self.finished = True
        else:
            tree.expand(node, None, None, None, None)

    def outcome(self) -> Tuple[bool, Optional[Route], int]:
        route = self.tree.best_route() if self.tree.succ else None
        return self.tree.succ, route, self.done_iters


def molstar(target_mol, target_mol_id, starting_mols, expand_fn, value_fn, iterations, viz=False, viz_dir=None,
            max_time=300, value_batch_fn=None) -> Tuple[bool, Optional[Route], int]:
    search = MolStarSearch(target_mol, starting_mols, value_fn, iterations, max_time, value_batch_fn)
    while True:
        node = search.select()
        if node is None:
            break
        search.apply(expand_fn(node.mol))
    return search.outcome()


def molstar_many(target_mols: Sequence[str], starting_mols, expand_batch_fn, value_fn, iterations, max_time=300,
                 value_batch_fn=None, clock=None) -> List[Tuple[bool, Optional[Route], int]]:
    """Independent A* searches advanced in lock step: each round every unfinished search nominates its best open
    molecule and ``expand_batch_fn([(search_index, mol), ...]) -> [result, ...]`` expands them together.  Every search
    sees exactly the expansions and estimates a solo ``molstar`` would, so with deterministic callbacks the routes are the
    same; ``max_time`` is measured on the shared wall clock, read once per round through ``clock`` (default ``time.time``; a
    replicated run passes a clock all ranks agree on).  With ``value_batch_fn`` the new tree nodes of ALL searches of a
    round are evaluated in ONE call (the targets themselves in one call before the first round): ~16 x 100 prompts per LLM
    value forward of BASELINE configs[2] instead of 100."""
    known = starting_mols if isinstance(starting_mols, (set, frozenset)) else set(starting_mols)
    roots = [None] * len(target_mols)
    if value_batch_fn is not None and len(target_mols) > 0:
        roots = [float(v) for v in value_batch_fn([(t, None) for t in target_mols])]
    searches = [MolStarSearch(t, known, value_fn, iterations, max_time, value_batch_fn, r) for t, r in zip(target_mols, roots)]
    if clock is not None:
        t0 = clock()
        for s in searches:
            s.t0 = t0
    while True:
        now = clock() if clock is not None else time.time()
        picks = [(i, s.select(now)) for i, s in enumerate(searches)]
        picks = [(i, n) for i, n in picks if n is not None]
        if not picks:
            break
        results = expand_batch_fn([(i, n.mol) for i, n in picks])
        assert len(results) == len(picks)
        if value_batch_fn is None:
            for (i, _), res in zip(picks, results):
                searches[i].apply(res)
            continue
        requests = [searches[i].value_requests(res) for (i, _), res in zip(picks, results)]
        flat = [r for reqs in requests for r in reqs]
        values = list(value_batch_fn(flat)) if flat else []
        assert len(values) == len(flat)
        lo = 0
        for (i, _), res, reqs in zip(picks, results, requests):
            searches[i].apply(res, values[lo:lo + len(reqs)])
            lo += len(reqs)
    return [s.outcome() for s in searches]
