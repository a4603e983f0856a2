"""One rank of the two-rank `main.py eval` dry run (tests/test_eval_gpu.py): installs the rdkit / rdchiral double for the life of the process
(so that graph -> SMILES, SMILES -> graph, validity and template application run the product's own code paths), then runs the eval driver."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tests import fake_rdkit  # noqa: E402

fake_rdkit.install_global(template_outcomes=fake_rdkit.split_template_runner)

import torch  # noqa: E402

from llamole_amd import eval as ev  # noqa: E402
from llamole_amd import molecule_utils  # noqa: E402

# A random-init denoiser samples junk graphs (atoms of valence 1 with three bonds, '*' in the middle of a chain): after the repair loop most
# of them fail the polymer check and come out as None -- a legitimate outcome, but it would leave the retrosynthesis tails idle.  The REAL
# graph_to_smiles runs on every generated graph (counted below); where it returns None the worker substitutes a small valid molecule so that
# check_valid, smiles_to_graph, the template runner and the A* search run on parseable molecules too.
_real_g2s = molecule_utils.graph_to_smiles
COUNTS = {"graphs": 0, "valid": 0}


def _counting_g2s(molecule_list, atom_decoder):
    out = _real_g2s(molecule_list, atom_decoder)
    COUNTS["graphs"] += len(out)
    COUNTS["valid"] += sum(o is not None for o in out)
    return [o if o is not None else f"C;C;{'NOS'[i % 3]};C|0-1:1,1-2:1,2-3:1" for i, o in enumerate(out)]


molecule_utils.graph_to_smiles = _counting_g2s

if __name__ == "__main__":
    torch.manual_seed(int(os.environ.get("RANK", "0")))
    out = ev.run_eval(sys.argv[1], overrides={"retro_iterations": 3, "retro_max_planning_time": 20})
    if int(os.environ.get("RANK", "0")) == 0:
        print("EVAL_STATS " + json.dumps({"n_results": len(out["results"]), "stats": out["stats"], "chemistry": COUNTS}))
