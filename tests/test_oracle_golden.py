"""CPU: the oracle restatement (oracle/rfn_oracle.py) against the golden vectors that oracle/make_golden.py
captured from the reference itself.  This is what pins the oracle (the reference ships no tests)."""
import numpy as np
import pytest
import torch

from conftest import load_case


def maxerr(a, b):
    return float((torch.as_tensor(a).double() - torch.as_tensor(b).double()).abs().max())


@pytest.mark.parametrize('name', ['tiny0', 'tiny1', 'tinymax', 'odd', 'mid'])
def test_forward_loss_and_every_gradient(name):
    from oracle import rfn_oracle as O
    cfg, spec, P, batch, gold = load_case(name)
    fc, att, labels, masks, top = batch
    log_prob, reason = O.forward(cfg, P, fc, att, labels)
    assert tuple(log_prob.shape) == tuple(gold['log_prob_shape'])
    assert maxerr(log_prob, gold['log_prob']) < 2e-5
    for j, r in enumerate(reason):
        assert maxerr(r, gold['reason_pred_%d' % j]) < 2e-5
    loss, grads = O.train_step_loss_and_grads(cfg, P, fc, att, labels, masks, top, 1.0)
    assert abs(float(loss.detach()) - float(gold['xe_loss'])) < 1e-4
    for k, g in grads.items():
        gn = float(gold['gradnorm/' + k])
        assert abs(float(g.double().norm()) - gn) <= 1e-6 + 1e-3 * gn, k
        if name.startswith('tiny'):
            assert maxerr(g, gold['grad/' + k]) < 2e-5 + 1e-4 * float(np.abs(gold['grad/' + k]).max()), k
    cfg.use_label_smoothing = 1
    ls = O.xe_criterion(cfg, log_prob, labels[:, 1:], masks[:, 1:], reason, top, 1.0)
    assert abs(float(ls) - float(gold['xe_loss_ls'])) < 1e-4


@pytest.mark.parametrize('name', ['tiny0', 'tinymax', 'mid'])
def test_training_mode_with_the_references_own_dropout_masks(name):
    """Dropout-mode pin: the reference was run in train() mode with drop_prob_fusion / _reason / _lm =
    0.1 / 0.2 / 0.3 and every nn.Dropout call's keep mask was captured by a forward hook.  The oracle with those masks
    must give the reference's log-probs, reason heads, loss and every gradient: which probability each stage uses, that
    every consumer sees the POST-dropout h while c is never dropped, the 1 / (1 - p) scale."""
    from conftest import load_drop_case
    from oracle import rfn_oracle as O
    cfg, spec, P, batch, gold, drop = load_drop_case(name)
    fc, att, labels, masks, top = batch
    log_prob, reason = O.forward(cfg, P, fc, att, labels, drop=drop)
    assert maxerr(log_prob, gold['log_prob']) < 2e-5
    for j, r in enumerate(reason):
        assert maxerr(r, gold['reason_pred_%d' % j]) < 2e-5
    loss, grads = O.train_step_loss_and_grads(cfg, P, fc, att, labels, masks, top, 1.0, drop=drop)
    assert abs(float(loss.detach()) - float(gold['xe_loss'])) < 1e-4
    for k, g in grads.items():
        gn = float(gold['gradnorm/' + k])
        assert abs(float(g.double().norm()) - gn) <= 1e-6 + 1e-3 * gn, k
        if name.startswith('tiny'):
            assert maxerr(g, gold['grad/' + k]) < 2e-5 + 1e-4 * float(np.abs(gold['grad/' + k]).max()), k
    # each stage's keep rate is its own probability's (not a shared one): 0.9 / 0.8 / 0.7 within sampling noise
    for key, p_ in (('keep_fusion', 0.1), ('keep_review', 0.2), ('keep_decoder', 0.3)):
        n = gold[key].size
        assert abs(float(gold[key].mean()) - (1 - p_)) < 4 * (p_ * (1 - p_) / n) ** 0.5 + 1e-3, key
    # a wrong wiring is visible: the eval-mode pass, or the masks with c dropped too, are far from the reference
    assert maxerr(O.forward(cfg, P, fc, att, labels)[0], gold['log_prob']) > 1e-3


def test_early_break_on_all_zero_column():
    """tiny0 has captions of <= 3 words with seq_length 5: the loop must stop at the first all-zero column
    (misc/RecurrentFusionModel.py:274), so fewer than seq_length+1 steps are produced."""
    cfg, spec, P, batch, gold = load_case('tiny0')
    assert int(gold['log_prob_shape'][1]) < cfg.seq_length + 2
    # steps = columns 0 .. last non-zero column (the BOS column 0 is always fed)
    assert int(gold['log_prob_shape'][1]) == int((batch[2] != 0).any(0).nonzero().max()) + 1


def test_adam_step_where_well_conditioned():
    from oracle import rfn_oracle as O
    cfg, spec, P, batch, gold = load_case('tiny1')
    loss, grads = O.train_step_loss_and_grads(cfg, P, *batch, 1.0)
    P2 = {k: v.clone() for k, v in P.items()}
    O.clip_and_adam(P2, grads, {}, lr=5e-4, weight_decay=1e-5, grad_clip=1.0)
    for k in P2:
        sel = torch.from_numpy(np.abs(gold['grad/' + k]) > 1e-5)
        if bool(sel.any()):
            assert maxerr(P2[k][sel], torch.from_numpy(gold['stepped/' + k])[sel]) < 2e-6, k


@pytest.mark.parametrize('name', ['tiny0', 'tiny1', 'tinymax', 'odd', 'mid', 'c2', 'c3'])
def test_greedy_sample(name):
    from oracle import rfn_oracle as O
    cfg, spec, P, batch, gold = load_case(name)
    fc, att = batch[0], batch[1]
    with torch.no_grad():
        seq, seq_lp, lp_all, _ = O.sample_greedy(cfg, P, fc, att)
    assert torch.equal(seq, torch.from_numpy(gold['greedy_seq']))
    assert maxerr(seq_lp, gold['greedy_seq_logprobs']) < 2e-5
    assert tuple(lp_all.shape) == tuple(gold['greedy_logprobs_all_shape'])


@pytest.mark.parametrize('name', ['c2'])
def test_shape_true_forward(name):
    from oracle import rfn_oracle as O
    cfg, spec, P, batch, gold = load_case(name)
    fc, att, labels, masks, top = batch
    with torch.no_grad():
        lp, reason = O.forward(cfg, P, fc, att, labels)
    idx = torch.from_numpy(gold['log_prob_top5_idx'])
    assert maxerr(lp.gather(2, idx), gold['log_prob_top5_val']) < 2e-5
    loss = O.xe_criterion(cfg, lp, labels[:, 1:], masks[:, 1:], reason, top, 1.0)
    assert abs(float(loss.detach()) - float(gold['xe_loss'])) < 1e-3


@pytest.mark.parametrize('name', ['tiny0', 'tinymax', 'odd', 'mid'])
def test_rl_replay(name):
    from oracle import rfn_oracle as O
    cfg, spec, P, batch, gold = load_case(name)
    fc, att, labels, masks, top = batch
    raw = torch.from_numpy(gold['rl_raw_ids'])
    with torch.no_grad():
        seq, seq_lp, lp_all, reason = O.sample_greedy(cfg, P, fc, att, force_ids=raw)
    assert torch.equal(seq, torch.from_numpy(gold['rl_seq']))
    loss = O.rl_criterion(cfg, seq_lp, seq, torch.from_numpy(gold['rl_reward']), lp_all, 0.01, reason, top, 1.0)
    assert abs(float(loss.detach()) - float(gold['rl_loss'])) < 1e-4


@pytest.mark.parametrize('name', ['tiny0', 'tinymax', 'odd', 'mid'])
def test_beam_search(name):
    from oracle import rfn_oracle as O
    cfg, spec, P, batch, gold = load_case(name)
    nb = gold['beam_seq'].shape[0]
    with torch.no_grad():
        out = O.sample_beam(cfg, P, [f[:nb] for f in batch[0]], [a[:nb] for a in batch[1]], int(gold['beam_size']))
    assert torch.equal(out[0], torch.from_numpy(gold['beam_seq']))
    assert maxerr(out[1], gold['beam_seq_logprobs']) < 2e-5
    for k in range(nb):
        assert torch.equal(out[2][k], torch.from_numpy(gold['beam_top_seq_%d' % k]))
        # cumulative scores are non-increasing in the sorted done list
        p = np.array(out[3][k])
        assert np.all(np.diff(p) <= 1e-6)


def test_single_cells():
    """a1 / a2 / a4 / a5 on the inputs the reference's own sub-modules were fed."""
    from oracle import rfn_oracle as O
    cfg, spec, P, batch, gold = load_case('mid')
    M, R = len(cfg.feat_array_info), cfg.rnn_size
    t = lambda k: torch.from_numpy(gold[k])  # noqa: E731
    h, c = t('cell_a2_h'), t('cell_a2_c')
    oh, oc, aux = O.fusion_cell(t('cell_a2_H'), batch[1][M - 1], h, c, P, 'review_steps_individual.1.lstm.%d.' % (M - 1), R)
    assert maxerr(oh, gold['cell_a2_out_h']) < 1e-5 and maxerr(oc, gold['cell_a2_out_c']) < 1e-5
    assert maxerr(aux['z'], gold['cell_a1_z']) < 1e-5 and maxerr(aux['alpha'], gold['cell_a1_alpha']) < 1e-6
    assert abs(float(aux['alpha'].sum(1).mean()) - 1.0) < 1e-6
    th = [t('cell_a4_thoughts_%d' % i) for i in range(M)]
    oh4, oc4, _ = O.review_cell(th, h, c, P, 2, R)
    assert maxerr(oh4, gold['cell_a4_out_h']) < 1e-5 and maxerr(oc4, gold['cell_a4_out_c']) < 1e-5
    oh5, oc5, _ = O.decoder_cell(t('cell_a5_xt'), t('cell_a5_comb'), h, c, P, R)
    assert maxerr(oh5, gold['cell_a5_out_h']) < 1e-5 and maxerr(oc5, gold['cell_a5_out_c']) < 1e-5


def test_multilabel_margin_written_out():
    """The explicit definition used to restate nn.MultiLabelMarginLoss, incl. no-target and duplicate rows."""
    import torch.nn.functional as F
    from oracle import rfn_oracle as O
    g = torch.Generator().manual_seed(0)
    pred = torch.randn(5, 12, generator=g)
    tgt = -torch.ones(5, 12, dtype=torch.long)
    tgt[0, :3] = torch.tensor([1, 5, 7])
    tgt[1, :2] = torch.tensor([0, 11])
    tgt[3, :1] = torch.tensor([4])
    assert abs(float(O.multilabel_margin(pred, tgt)) - float(F.multilabel_margin_loss(pred, tgt))) < 1e-6
