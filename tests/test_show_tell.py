"""BASELINE config 1: ShowTellModel single-encoder greedy decode on the CPU (B=4, 2048-d features, seq_len 16) through
`models.setup(opt)` -- plumbing, no GPU -- against outputs captured from the reference's own ShowTellModel
(oracle/make_golden.py generate_showtell; misc/ShowTellModel.py:10-240, misc/utils.py:252-282)."""
import os

import numpy as np
import torch

from conftest import GOLDEN_DIR


def _case():
    from oracle import make_golden as G
    import recurrent_fusion_network_amd as R
    gold = np.load(os.path.join(GOLDEN_DIR, 'showtell.npz'))
    cfg = G.showtell_cfg()
    model = R.setup(cfg)
    assert isinstance(model, R.ShowTellModel)
    assert sorted(model.state_dict()) == list(gold['state_dict_keys'])          # reference checkpoint keys
    W = G.showtell_weights(model, G.SHOWTELL['seed'])
    assert abs(G.digest([W[k] for k in sorted(W)]) - float(gold['weights_digest'])) <= 1e-6 * abs(float(gold['weights_digest']))
    model.load_state_dict(W)
    fc, labels, masks = G.showtell_batch()
    assert np.array_equal(labels.numpy(), gold['labels'])
    return R, cfg, model.eval(), fc, labels, masks, gold


def test_show_tell_forward_and_criterion_match_the_reference():
    R, cfg, model, fc, labels, masks, gold = _case()
    with torch.no_grad():
        lp = model(fc, None, labels)
    assert tuple(lp.shape) == tuple(gold['log_prob_shape'])                     # early break at the first all-zero column
    idx = torch.from_numpy(gold['log_prob_top5_idx'])
    assert float((lp.gather(2, idx) - torch.from_numpy(gold['log_prob_top5_val'])).abs().max()) < 1e-5
    crit = R.LanguageModelCriterion(cfg)
    assert abs(float(crit(lp, labels[:, 1:], masks[:, 1:])) - float(gold['xe_loss'])) < 1e-4
    cfg.use_label_smoothing = 1
    assert abs(float(R.LanguageModelCriterion(cfg)(lp, labels[:, 1:], masks[:, 1:])) - float(gold['xe_loss_ls'])) < 1e-4


def test_show_tell_greedy_decode_ids_exact():
    R, cfg, model, fc, labels, masks, gold = _case()
    with torch.no_grad():
        seq, seq_lp, lp_all = model.sample(fc, None, {'sample_max': 1})
    assert torch.equal(seq, torch.from_numpy(gold['greedy_seq']))
    assert float((seq_lp - torch.from_numpy(gold['greedy_seq_logprobs'])).abs().max()) < 1e-5
    assert tuple(lp_all.shape) == tuple(gold['greedy_logprobs_all_shape'])
    # a train step runs (forward -> criterion -> backward), gradients reach every parameter
    model.train()
    loss = R.LanguageModelCriterion(cfg)(model(fc, None, labels), labels[:, 1:], masks[:, 1:])
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())


def test_show_tell_beam_search_matches_the_reference():
    """misc/ShowTellModel.py:95-185 through sample(beam_size=3): the reference's own rules (only beam 0 at the first merge,
    column-major candidates, stable sort, a beam that emitted END keeps competing) -- on the seeded weights, where no beam
    ends early, and with the END logit raised, where the done lists grow to 34-46 entries per image."""
    from oracle import make_golden as G
    R, cfg, model, fc, labels, masks, gold = _case()
    W = {k: v.clone() for k, v in model.state_dict().items()}
    for tag, bias in (('beam3', 0.0), ('beam3e', G.SHOWTELL_END_BIAS)):
        Wb = dict(W)
        Wb['logit.bias'] = W['logit.bias'].clone()
        Wb['logit.bias'][0] += bias
        model.load_state_dict(Wb)
        seq, seq_lp = model.sample(fc, None, {'beam_size': 3})
        assert torch.equal(seq, torch.from_numpy(gold[tag + '_seq'])), tag
        assert float((seq_lp - torch.from_numpy(gold[tag + '_seq_logprobs'])).abs().max()) < 1e-5
        counts = [len(d) for d in model.done_beams]
        assert counts == list(gold[tag + '_done_counts']), tag
        for k, d in enumerate(model.done_beams):
            assert np.allclose([b['p'] for b in d], gold[tag + '_done_p'][k][:len(d)], atol=1e-5)
            assert all(np.array_equal(b['seq'].numpy(), gold[tag + '_done_seq'][k][j]) for j, b in enumerate(d))
            assert all(d[j]['p'] >= d[j + 1]['p'] for j in range(len(d) - 1))       # best first
    assert max(gold['beam3e_done_counts']) > 3          # the END rule was exercised
