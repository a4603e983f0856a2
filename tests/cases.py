"""Shared definitions of the golden-fixture cases (used by the generator and the tests)."""
import os

import numpy as np

from llamole_amd import synth

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

DIT_CASES = {
    # name: (max_node, H, L, heads, T, guide, batch, seed)
    "dit_n32_h128": (32, 128, 2, 4, 50, 2.0, 8, 0),     # hd=32; BASELINE cfg-1 shape (B=8, N=32, T=50)
    "dit_n50_h256": (50, 256, 2, 4, 12, 1.5, 3, 1),     # hd=64; ragged n_nodes; N not a power of 2
}

GIN_CASES = {
    # name: (num_layer, H, out_dim, n_graphs, seed)
    "gin_l3_h64": (3, 64, 1000, 6, 0),
    "gin_l5_h128": (5, 128, 2048, 16, 1),
}


def dit_case(name):
    N, H, L, heads, T, guide, B, seed = DIT_CASES[name]
    cfg = synth.make_dit_config(H, L, heads, T, guide)
    meta = synth.make_data_meta(N, seed)
    sd = synth.make_dit_weights(cfg, N, seed)
    return cfg, meta, sd, B, seed


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))


def fake_template_runner(template: str, smiles: str):
    """Deterministic stand-in for rdchiralRunText used to pin the host merge logic of sample_templates."""
    i = int(template[1:])
    if i % 7 == 0:
        return []
    if i % 11 == 0:
        raise RuntimeError("template does not apply")
    outs = [f"R{i % 4}.A{i % 3}", f"B{i % 6}"]
    if i % 2 == 0:
        outs.append(f"A{i % 3}.R{i % 4}")      # same reactant set, different order -> merged
    if i % 3 == 0:
        outs.append(f"C{i % 5}.C{i % 5}.D")
    return outs
