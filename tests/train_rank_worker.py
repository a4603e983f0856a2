"""One rank of the two-rank `main.py train` dry run (tests/test_train_gpu.py): SMILES -> graph comes from the ring-graph maker of
tests/host_fakes.py (rdkit is in neither image), everything else is the product's own path."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from llamole_amd import train as tr  # noqa: E402
from llamole_amd.graph_data import GraphData  # noqa: E402
from llamole_amd.modeling_llamole import GraphLLMForCausalMLM  # noqa: E402
from tests.host_fakes import fake_smiles_to_graph  # noqa: E402

_to_graph = fake_smiles_to_graph(GraphData)
GraphLLMForCausalMLM.smiles_to_graph = lambda self, s: _to_graph(s)

if __name__ == "__main__":
    torch.manual_seed(0)
    out = tr.run_train(sys.argv[1])
    if int(os.environ.get("RANK", "0")) == 0:
        print("TRAIN_LOG " + json.dumps({"losses": [r["loss"] for r in out["log"]], "lora_modules": out["lora_modules"]}))
