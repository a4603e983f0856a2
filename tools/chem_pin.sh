#!/bin/bash
# Pin the host chemistry tails to the real rdkit / rdchiral and to the reference's own code.
#
# Neither package exists in the build image or on the GPU box (no network there), so this script cannot run in this repository's own
# rounds; it is the ONE command that turns a15 / f3 from "partial" to "pinned" on any machine with network access and a checkout of
# the reference (liugangcode/Llamole):
#
#     tools/chem_pin.sh /path/to/Llamole [venv_dir]
#
# What it does:
#   1. creates a virtualenv with the reference's pins (requirements.txt:21-22): rdkit==2023.9.6, rdchiral==1.1.0, + numpy, pandas, CPU torch,
#      pytest;
#   2. runs tests/test_chemistry_real.py -- known-answer tests of llamole_amd/molecule_utils.py, GraphPredictor.smiles_to_fp /
#      merge_template_outcomes and smiles_to_graph against real chemistry (their expected strings were written from the reference's
#      semantics, not from a run: a failure here is a finding about the restatement OR about the expectation);
#   3. runs tools/chem_pin_compare.py -- imports the reference's graph_decoder/molecule_utils.graph_to_smiles and
#      graph_predictor/model.GraphPredictor.sample_templates BESIDE this repository's and asserts equal results on 200 seeded integer graphs
#      and 40 seeded (product, template list) cases.
# Exit code 0 = pinned.  Nothing under llamole_amd/ imports either script; no GPU is needed (the HIP library is not loaded).
set -euo pipefail
REF=${1:?usage: tools/chem_pin.sh /path/to/Llamole [venv_dir]}
VENV=${2:-.venv_chem_pin}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
[ -f "$REF/src/model/graph_decoder/molecule_utils.py" ] || { echo "not a Llamole checkout: $REF" >&2; exit 2; }
python3 -m venv "$VENV"
# shellcheck disable=SC1091
. "$VENV/bin/activate"
pip install --upgrade pip
pip install "rdkit==2023.9.6" "rdchiral==1.1.0" "numpy<2" pandas pyyaml pytest
pip install torch --index-url https://download.pytorch.org/whl/cpu
cd "$ROOT"
python -m pytest tests/test_chemistry_real.py -q
python tools/chem_pin_compare.py --reference "$REF"
echo "chem_pin: rows a15 / f3 are pinned to rdkit $(python -c 'import rdkit; print(rdkit.__version__)') and to the reference at $REF"
