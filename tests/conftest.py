import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(GOLDEN, "reference_compute_outputs.npz"))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement (test infrastructure; builds oracle/libtapir_oracle.so with gcc on first use)."""
    from oracle import oracle as orc
    orc.lib()
    return orc


@pytest.fixture(scope="session")
def chr1_918(golden_dir):
    """The reference's bundled locus + tree, as the engine wants them (tree already / correction)."""
    import json
    from tapir_amd import newick, nexus
    from tapir_amd.compute import correct_tree
    names, states = nexus.read_states(os.path.join(golden_dir, "chr1_918.nex"))
    root = newick.read_tree(os.path.join(golden_dir, "Euteleost.tree"))
    depth, factor = correct_tree(root)
    parent, blen, leaf = newick.to_arrays(root, names)
    kat = json.load(open(os.path.join(golden_dir, "chr1_918_phydesign_rates.json")))
    pi = np.array(kat["freqs_ACGT"])
    exch = np.array([kat[k] for k in ("AC", "AG", "AT", "CG", "CT", "GT")])
    return dict(names=names, states=states, parent=parent, blen=blen, leaf=leaf, depth=depth, factor=factor,
                pi=pi, exch=exch, kat=kat)
