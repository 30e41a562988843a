"""Training-mode (dropout > 0) parity of the whole path.

The chain of evidence: (1) tests/test_oracle_golden.py pins the oracle WITH explicit dropout masks to the reference run in
train() mode (masks captured from its own nn.Dropout layers, tests/golden/*_drop.npz); (2) here the HIP path runs in
training mode with drop_prob_fusion / _reason / _lm = 0.1 / 0.2 / 0.3, its Philox masks are read back through
rfn_dropout_mask (include/rfn.h documents the (seed, call-site offset, element index) convention) and handed to that same
oracle: log-probs <= 1e-3, reason heads, loss and EVERY gradient must agree.  That checks on the product which
probability each stage uses, that the reason heads / logit layer / concatenated H / thought vectors see the POST-dropout
h while c is never dropped (misc/RecurrentFusionModel.py:68-73, misc/LSTMSoftMultiAttentionFeatArrayNoInputCore.py:67-72,
misc/LSTMSoftAttentionCore.py:96-101), the 1 / (1 - p) scale, and that backward regenerates the forward's masks."""
import pytest
import torch

from conftest import load_drop_case
from test_model_gpu import LOGP_TOL, build, maxerr, to_dev

pytestmark = pytest.mark.gpu

OFF_STAGE2, OFF_DECODER = 1 << 20, 1 << 21        # RFN_DROP_OFFSET_STAGE2 / _DECODER (include/rfn.h)


def product_masks(cfg, B, S, seed, dev):
    """The keep masks librfn_hip.so applies for `seed`, as an oracle `drop` object."""
    import recurrent_fusion_network_amd._native as N
    from oracle import rfn_oracle as O
    M, R = len(cfg.feat_array_info), cfg.rnn_size

    def site(offset, p):
        keep = torch.empty(B, R, device=dev)
        N.check(N.lib.rfn_dropout_mask(seed, offset, B * R, p, keep.data_ptr(), N.stream_ptr()), 'rfn_dropout_mask')
        return keep.cpu()

    fusion = [[site(t * M + i, cfg.drop_prob_fusion) for i in range(M)] for t in range(cfg.num_review_steps_0)]
    review = [site(OFF_STAGE2 + t, cfg.drop_prob_reason) for t in range(cfg.num_review_steps)]
    decoder = [site(OFF_DECODER + s, cfg.drop_prob_lm) for s in range(S)]
    return O.make_drop(cfg, fusion, review, decoder)


@pytest.mark.parametrize('name', ['tiny0', 'tinymax', 'mid'])
def test_training_mode_against_the_oracle_with_the_products_own_masks(dev, name):
    import recurrent_fusion_network_amd as R
    from oracle import rfn_oracle as O
    from recurrent_fusion_network_amd.fusion_model import _fresh_seed
    cfg, spec, P, batch, gold, ref_drop = load_drop_case(name)
    fc, att, labels, masks, top = batch
    model = build(cfg, P, dev, train=True)
    torch.manual_seed(17)
    seed = _fresh_seed()                       # the seed forward() will draw from torch's CPU generator
    torch.manual_seed(17)
    d = to_dev(batch, dev)
    log_prob, reason = model(d[0], d[1], d[2])
    crit = R.ReviewNetEnsembleCriterion(cfg)
    loss = crit(log_prob, d[2][:, 1:], d[3][:, 1:], reason, d[4], 1.0)
    loss.backward()

    S = log_prob.size(1)
    drop = product_masks(cfg, labels.size(0), S, seed, dev)
    # the masks are Bernoulli(1 - p) with each stage's own p
    for key, p_ in (('fusion', cfg.drop_prob_fusion), ('review', cfg.drop_prob_reason), ('decoder', cfg.drop_prob_lm)):
        flat = torch.cat([m.reshape(-1) for row in drop[key] for m in (row if isinstance(row, list) else [row])])
        assert abs(float(flat.mean()) - (1 - p_)) < 4 * (p_ * (1 - p_) / flat.numel()) ** 0.5 + 1e-3, key
    o_lp, o_reason = O.forward(cfg, P, fc, att, labels, drop=drop)
    assert tuple(o_lp.shape) == tuple(log_prob.shape)
    assert maxerr(log_prob, o_lp) < LOGP_TOL
    for a, b in zip(reason, o_reason):
        assert maxerr(a, b) < LOGP_TOL
    o_loss, o_grads = O.train_step_loss_and_grads(cfg, P, fc, att, labels, masks, top, 1.0, drop=drop)
    assert abs(float(loss.detach()) - float(o_loss)) < 1e-4 * max(1.0, abs(float(o_loss)))
    named = dict(model.named_parameters())
    for k, g in o_grads.items():
        assert maxerr(named[k].grad, g) < 1e-5 + 1e-3 * float(g.abs().max()), k
    # not vacuous: the eval-mode oracle, and the oracle with the REFERENCE run's masks, are far from this pass
    assert maxerr(log_prob, O.forward(cfg, P, fc, att, labels)[0]) > 10 * LOGP_TOL
    assert maxerr(log_prob, O.forward(cfg, P, fc, att, labels, drop=ref_drop)[0]) > 10 * LOGP_TOL


def test_free_running_sample_under_dropout_against_the_oracle(dev):
    """sample() in train() mode (train_rl.py:160 samples with dropout on): the greedy free-running decode applies the
    decoder masks of (seed, step); replaying the product's masks through the oracle's teacher-forced pass on the tokens it
    fed gives the same log-probs."""
    from oracle import rfn_oracle as O
    from recurrent_fusion_network_amd.fusion_model import _fresh_seed
    cfg, spec, P, batch, gold, _ = load_drop_case('mid')
    fc, att, labels, masks, top = batch
    model = build(cfg, P, dev, train=True)
    d = to_dev(batch, dev)
    torch.manual_seed(23)
    seed = _fresh_seed()
    torch.manual_seed(23)
    with torch.no_grad():
        seq, seq_lp, lp_all, reason = model.sample(d[0], d[1], {'sample_max': 1})
    T = lp_all.size(1)
    drop = product_masks(cfg, labels.size(0), T, seed, dev)
    fed = torch.zeros(labels.size(0), T, dtype=torch.long)
    # tokens fed at steps 1.. are the UNMASKED argmax of the previous distribution (misc/RecurrentFusionModel.py:620,637)
    fed[:, 1:] = lp_all[:, :T - 1].argmax(2).cpu()
    fed[:, 1:][fed[:, 1:] == 0] = 0
    o_lp, o_reason = O.forward(cfg, P, fc, att, torch.cat([fed, torch.zeros(fed.size(0), 1, dtype=torch.long)], 1), drop=drop)
    # O.forward stops at the first all-zero column after column 0: compare the steps both produced
    n = min(o_lp.size(1), T)
    assert n >= 2 and maxerr(lp_all[:, :n], o_lp[:, :n]) < LOGP_TOL
    for a, b in zip(reason, o_reason):
        assert maxerr(a, b) < LOGP_TOL
