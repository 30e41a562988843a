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


def load_drop_case(name):
    """Training-mode tier `<name>_drop` (oracle/make_golden.py generate_dropout): the base tier's weights and inputs
    with drop_prob_fusion / _reason / _lm = 0.1 / 0.2 / 0.3, plus the keep masks the reference's own nn.Dropout layers
    drew and the results it produced with them.  -> (cfg, spec, P, batch, gold, drop) with drop = O.make_drop(...)."""
    from oracle import make_golden as G
    from oracle import rfn_oracle as O
    spec = dict(G.CONFIGS[name])
    spec['extra'] = dict(spec.get('extra', {}), **G.DROP_PROBS)
    cfg = G.cfg_of(spec)
    P = O.seeded_params(cfg, spec['seed'])
    fc, att, labels, masks, top = G.batch_of(cfg, spec)
    gold = np.load(os.path.join(GOLDEN_DIR, name + '_drop.npz'))
    assert abs(G.digest([P[k] for k in sorted(P)]) - float(gold['weights_digest'])) <= 1e-6 * abs(float(gold['weights_digest']))
    assert abs(G.digest(fc + att) - float(gold['inputs_digest'])) <= 1e-6 * abs(float(gold['inputs_digest']))
    assert np.array_equal(labels.numpy(), gold['labels'])
    assert np.allclose(gold['drop_probs'], [cfg.drop_prob_fusion, cfg.drop_prob_reason, cfg.drop_prob_lm])
    kf, kr, kd = (torch.from_numpy(gold[k]) for k in ('keep_fusion', 'keep_review', 'keep_decoder'))
    drop = O.make_drop(cfg, [[kf[t, i] for i in range(kf.size(1))] for t in range(kf.size(0))], list(kr), list(kd))
    return cfg, spec, P, (fc, att, labels, masks, top), gold, drop


@pytest.fixture(scope='session')
def dev():
    return torch.device('cuda:0')
