import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, 'tests', 'golden')


def _ensure_library():
    """The product library is a build artefact (git-ignored): a fresh checkout builds it once, the same way
    __graft_entry__.build() does.  A failing build fails the run -- there is no fallback to test instead."""
    so = os.path.join(ROOT, 'recurrent_fusion_network_amd', 'librfn_hip.so')
    if not os.path.exists(so):
        import subprocess
        subprocess.run(['make', '-C', os.path.join(ROOT, 'recurrent_fusion_network_amd', 'csrc'), '-j4'], check=True,
                       stdout=subprocess.DEVNULL)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    _ensure_library()


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


def load_case(name):
    """(cfg, spec, params, batch, golden) for one golden tier; weights and inputs are regenerated from the
    documented seeds and checked against the digests stored by oracle/make_golden.py."""
    from oracle import make_golden as G
    from oracle import rfn_oracle as O
    spec = G.CONFIGS[name]
    cfg = G.cfg_of(spec)
    P = O.seeded_params(cfg, spec['seed'])
    fc, att, labels, masks, top = G.batch_of(cfg, spec)
    gold = np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
    assert abs(G.digest([P[k] for k in sorted(P)]) - float(gold['weights_digest'])) <= 1e-6 * abs(
        float(gold['weights_digest'])), 'weight stream drifted from the golden run'
    assert abs(G.digest(fc + att) - float(gold['inputs_digest'])) <= 1e-6 * abs(
        float(gold['inputs_digest'])), 'input stream drifted from the golden run'
    assert np.array_equal(labels.numpy(), gold['labels'])
    return cfg, spec, P, (fc, att, labels, masks, top), gold


@pytest.fixture(scope='session')
def dev():
    return torch.device('cuda:0')
