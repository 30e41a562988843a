"""Whole-path parity on the GPU: RecurrentFusionModel (HIP) vs the golden vectors produced by the
reference and vs the CPU oracle on the same seeded inputs.

Tolerances (north star): log-probs / logits within 1e-3 absolute in fp32 (measured ~1e-5); greedy token
ids bit-exact; gradients within 1e-3 relative to each tensor's max |g| (+ a small absolute floor).
"""
import numpy as np
import pytest
import torch

from conftest import load_case

pytestmark = pytest.mark.gpu

LOGP_TOL = 1e-3


def build(cfg, P, dev, train=False):
    import recurrent_fusion_network_amd as R
    model = R.RecurrentFusionModel(cfg)
    model.load_state_dict(P)
    model = model.to(dev)
    model.train(train)
    return model


def to_dev(batch, dev):
    fc, att, labels, masks, top = batch
    return [t.to(dev) for t in fc], [t.to(dev) for t in att], labels.to(dev), masks.to(dev), top.to(dev)


def maxerr(a, b):
    return float((a.detach().double().cpu() - torch.as_tensor(b).double()).abs().max())


@pytest.mark.parametrize('name', ['tiny0', 'tiny1', 'tinymax', 'odd', 'mid'])
def test_forward_loss_grads_full(dev, name):
    import recurrent_fusion_network_amd as R
    from oracle import rfn_oracle as O
    cfg, spec, P, batch, gold = load_case(name)
    model = build(cfg, P, dev)
    fc, att, labels, masks, top = to_dev(batch, dev)
    log_prob, top_pred = model(fc, att, labels)
    assert tuple(log_prob.shape) == tuple(gold['log_prob_shape'])
    assert maxerr(log_prob, gold['log_prob']) < LOGP_TOL
    for j, r in enumerate(top_pred):
        assert maxerr(r, gold['reason_pred_%d' % j]) < LOGP_TOL
    crit = R.ReviewNetEnsembleCriterion(cfg)
    loss = crit(log_prob, labels[:, 1:], masks[:, 1:], top_pred, top, 1.0)
    assert abs(float(loss.detach()) - float(gold['xe_loss'])) < 1e-4 * max(1.0, abs(float(gold['xe_loss'])))
    loss.backward()
    o_loss, o_grads = O.train_step_loss_and_grads(cfg, P, *batch, 1.0)
    named = dict(model.named_parameters())
    worst = ('', 0.0)
    for k, g in o_grads.items():
        got = named[k].grad
        assert got is not None, k
        tol = 1e-5 + 1e-3 * float(g.abs().max())
        err = maxerr(got, g)
        if err / tol > worst[1]:
            worst = (k, err / tol)
        assert err < tol, (k, err, tol)
        assert abs(float(got.double().norm()) - float(gold['gradnorm/' + k])) <= 1e-5 + 2e-3 * float(gold['gradnorm/' + k])
    # label smoothing variant of the criterion
    cfg.use_label_smoothing = 1
    crit_ls = R.ReviewNetEnsembleCriterion(cfg)
    loss_ls = crit_ls(log_prob.detach(), labels[:, 1:], masks[:, 1:], [t.detach() for t in top_pred], top, 1.0)
    assert abs(float(loss_ls) - float(gold['xe_loss_ls'])) < 1e-4 * max(1.0, abs(float(gold['xe_loss_ls'])))


def _gemm_mode(model, gemm):
    """'bf16x3': the hoisted projections and their weight gradients on the bf16 matrix cores (RFN_GEMM_OPT_BF16X3), at any
    size, so that the small reference-generated tiers go through the plane GEMM too."""
    if gemm == 'bf16x3':
        import recurrent_fusion_network_amd._native as N
        model.gemm_flags |= N.GEMM_OPT_BF16X3 | N.GEMM_OPT_BF16X3_ANY_SIZE
    return model


@pytest.mark.parametrize('gemm', ['exact', 'bf16x3'])
@pytest.mark.parametrize('name', ['c2', 'c3'])
def test_forward_loss_grads_shape_true(dev, name, gemm):
    """Shape-true tiers (R=A=E=512, V+1=9488; c3: M=4, L=196, D=2048): goldens hold top-5 log-probs, target
    log-probs, loss, and per-parameter gradient norms + strided slices.  Both GEMM modes against the same
    reference-generated numbers, same tolerances."""
    import recurrent_fusion_network_amd as R
    cfg, spec, P, batch, gold = load_case(name)
    model = _gemm_mode(build(cfg, P, dev), gemm)
    fc, att, labels, masks, top = to_dev(batch, dev)
    log_prob, top_pred = model(fc, att, labels)
    assert tuple(log_prob.shape) == tuple(gold['log_prob_shape'])
    lp = log_prob.detach().cpu()
    idx = torch.from_numpy(gold['log_prob_top5_idx'])
    assert float((lp.gather(2, idx) - torch.from_numpy(gold['log_prob_top5_val'])).abs().max()) < LOGP_TOL
    tgt = labels.cpu()[:, 1:1 + lp.size(1)]
    assert float((lp.gather(2, tgt.unsqueeze(2)).squeeze(2) - torch.from_numpy(gold['log_prob_target'])).abs().max()) < LOGP_TOL
    # top-1 of every step agrees wherever the reference's own margin is not razor thin
    top2 = torch.from_numpy(gold['log_prob_top5_val'])
    safe = (top2[:, :, 0] - top2[:, :, 1]) > 1e-4
    assert torch.equal(lp.argmax(2)[safe], idx[:, :, 0][safe])
    for j, r in enumerate(top_pred):
        assert maxerr(r[:, :32], gold['reason_pred_%d' % j]) < LOGP_TOL
        assert float((r.detach().double().sum(1).cpu() - torch.from_numpy(gold['reason_pred_rowsum_%d' % j])).abs().max()) < 2e-2
    crit = R.ReviewNetEnsembleCriterion(cfg)
    loss = crit(log_prob, labels[:, 1:], masks[:, 1:], top_pred, top, 1.0)
    assert abs(float(loss.detach()) - float(gold['xe_loss'])) < 1e-4 * abs(float(gold['xe_loss']))
    loss.backward()
    for k, p in model.named_parameters():
        gn = float(gold['gradnorm/' + k])
        got = p.grad.detach().reshape(-1)
        assert abs(float(got.double().norm()) - gn) <= 1e-6 + 2e-3 * gn, (k, float(got.double().norm()), gn)
        stride = max(1, got.numel() // 16)
        sl = got[::stride][:16].cpu().double()
        ref = torch.from_numpy(gold['gradslice/' + k]).double()
        assert float((sl - ref).abs().max()) <= 1e-6 + 2e-3 * max(float(ref.abs().max()), gn / max(1.0, got.numel() ** 0.5)), k


@pytest.mark.parametrize('name,gemm', [(n, 'exact') for n in ('tiny0', 'tiny1', 'tinymax', 'odd', 'mid', 'c2', 'c3')] +
                         [('c2', 'bf16x3'), ('c3', 'bf16x3')])
def test_greedy_sample_ids_bit_exact(dev, name, gemm):
    cfg, spec, P, batch, gold = load_case(name)
    model = _gemm_mode(build(cfg, P, dev), gemm)
    fc, att, labels, masks, top = to_dev(batch, dev)
    with torch.no_grad():
        seq, seq_lp, lp_all, reason = model.sample(fc, att, {'sample_max': 1, 'beam_size': 1})
    ref_seq = torch.from_numpy(gold['greedy_seq'])
    margin = torch.from_numpy(gold['greedy_margin'])
    assert tuple(lp_all.shape) == tuple(gold['greedy_logprobs_all_shape'])
    assert tuple(seq.shape) == tuple(ref_seq.shape)
    assert torch.equal(seq.cpu(), ref_seq), 'greedy ids differ (smallest reference margin %.3g)' % float(margin.min())
    assert maxerr(seq_lp, gold['greedy_seq_logprobs']) < LOGP_TOL
    if 'greedy_logprobs_all' in gold:
        assert maxerr(lp_all, gold['greedy_logprobs_all']) < LOGP_TOL
    else:
        idx = torch.from_numpy(gold['greedy_top5_idx'])
        assert float((lp_all.cpu().gather(2, idx) - torch.from_numpy(gold['greedy_top5_val'])).abs().max()) < LOGP_TOL


@pytest.mark.parametrize('name', ['tiny0', 'odd', 'mid'])
def test_adam_step_matches_reference(dev, name):
    """One clamp + Adam step (train.py:162-163) through FusedClampAdam on the flat buffers."""
    import recurrent_fusion_network_amd as R
    from oracle import rfn_oracle as O
    cfg, spec, P, batch, gold = load_case(name)
    model = build(cfg, P, dev)
    opt = R.FusedClampAdam(model, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5, grad_clip=1.0)
    fc, att, labels, masks, top = to_dev(batch, dev)
    crit = R.ReviewNetEnsembleCriterion(cfg)
    opt.zero_grad()
    log_prob, top_pred = model(fc, att, labels)
    crit(log_prob, labels[:, 1:], masks[:, 1:], top_pred, top, 1.0).backward()
    opt.step()
    o_loss, o_grads = O.train_step_loss_and_grads(cfg, P, *batch, 1.0)
    named = dict(model.named_parameters())
    for k in P:
        # the first Adam step is lr*g/(|g|+eps): compare where |g| >> eps (see oracle/make_golden.py)
        sel = o_grads[k].abs() > 1e-5
        if not bool(sel.any()):
            continue
        if name == 'tiny0':
            want = torch.from_numpy(gold['stepped/' + k])
        else:
            want = O.clip_and_adam({k: P[k].clone()}, {k: o_grads[k]}, {}, lr=5e-4, weight_decay=1e-5)[k]
        got = named[k].detach().cpu()
        assert float((got[sel] - want[sel]).abs().max()) < 5e-6, k


@pytest.mark.parametrize('name', ['tiny0', 'tiny1', 'tinymax', 'odd', 'mid', 'c2'])
def test_rl_sample_replay_and_reward_criterion(dev, name):
    """train_rl.py:160-191: multinomial sample with grad (ids replayed from the reference's draw), reward
    criterion, backward."""
    import recurrent_fusion_network_amd as R
    cfg, spec, P, batch, gold = load_case(name)
    model = build(cfg, P, dev)
    fc, att, labels, masks, top = to_dev(batch, dev)
    raw = torch.from_numpy(gold['rl_raw_ids'])
    seq, seq_lp, lp_all, reason = model.sample(fc, att, {'sample_max': 0, 'force_ids': raw})
    assert torch.equal(seq.cpu(), torch.from_numpy(gold['rl_seq']))
    assert maxerr(seq_lp, gold['rl_seq_logprobs']) < LOGP_TOL
    crit = R.ReviewNetRewardCriterion(cfg)
    reward = torch.from_numpy(gold['rl_reward']).to(dev)
    loss = crit(seq_lp, seq, reward, lp_all, 0.01, reason, top, 1.0, None, cfg)
    assert abs(float(loss.detach()) - float(gold['rl_loss'])) < 1e-4 * max(1.0, abs(float(gold['rl_loss'])))
    loss.backward()
    for k, p in model.named_parameters():
        gn = float(gold['rl_gradnorm/' + k])
        assert abs(float(p.grad.double().norm()) - gn) <= 1e-5 + 3e-3 * gn, k


@pytest.mark.parametrize('name', ['tiny0', 'tinymax', 'odd', 'mid', 'c2'])
def test_beam_search_matches_reference(dev, name):
    cfg, spec, P, batch, gold = load_case(name)
    model = build(cfg, P, dev)
    fc, att, labels, masks, top = to_dev(batch, dev)
    nb = gold['beam_seq'].shape[0]
    beam = int(gold['beam_size'])
    out = model.sample([f[:nb] for f in fc], [a[:nb] for a in att], {'beam_size': beam})
    seq, seq_lp, top_seq, top_prob, reason = out
    assert torch.equal(seq.cpu(), torch.from_numpy(gold['beam_seq']))
    assert maxerr(seq_lp, gold['beam_seq_logprobs']) < LOGP_TOL
    for k in range(nb):
        assert torch.equal(top_seq[k], torch.from_numpy(gold['beam_top_seq_%d' % k]))
        assert np.allclose(np.array(top_prob[k]), gold['beam_top_prob_%d' % k], atol=1e-3)
    assert len(model.done_beams) == nb


def test_beam_search_is_independent_of_batch_composition(dev):
    """All images share one decoder batch; an image's result must not depend on its neighbours, and beam
    scores must be sorted (cumulative log-probs non-increasing)."""
    cfg, spec, P, batch, gold = load_case('mid')
    model = build(cfg, P, dev)
    fc, att, labels, masks, top = to_dev(batch, dev)
    together = model.sample(fc, att, {'beam_size': 5})
    done_together = [list(d) for d in model.done_beams]
    for k in (0, 3, 5):
        alone = model.sample([f[k:k + 1] for f in fc], [a[k:k + 1] for a in att], {'beam_size': 5})
        assert torch.equal(alone[0][0], together[0][k])
        assert maxerr(alone[1][0], together[1][k].cpu()) < 1e-5
        assert torch.equal(alone[2][0], together[2][k])
        p = np.array(together[3][k])
        assert np.all(np.diff(p) <= 1e-6) and len(p) == len(done_together[k])
    # greedy == best beam's first token when the beam is wide enough to contain the greedy path's prefix
    with torch.no_grad():
        g = model.sample(fc, att, {'sample_max': 1})[0]
    assert g.shape[0] == together[0].shape[0]


def test_dropout_training_mode_runs_and_is_seeded(dev):
    cfg, spec, P, batch, gold = load_case('tiny0')
    cfg.drop_prob_lm, cfg.drop_prob_reason, cfg.drop_prob_fusion = 0.3, 0.2, 0.1
    model = build(cfg, P, dev, train=True)
    fc, att, labels, masks, top = to_dev(batch, dev)
    torch.manual_seed(5)
    a, _ = model(fc, att, labels)
    torch.manual_seed(5)
    b, _ = model(fc, att, labels)
    torch.manual_seed(6)
    c, _ = model(fc, att, labels)
    assert torch.equal(a, b) and not torch.equal(a, c)
    a.sum().backward()
    assert all(torch.isfinite(p.grad).all() for p in model.parameters())
    model.eval()
    e, _ = model(fc, att, labels)
    assert maxerr(e, gold['log_prob']) < LOGP_TOL


def test_dedup_of_replicated_images_is_exact(dev):
    """dataloader.py:251-252 repeats every image seq_per_img times; running stages I/II once per image must give
    the same log-probs and the same gradients as the plain path on the replicated batch."""
    import recurrent_fusion_network_amd as R
    cfg, spec, P, batch, gold = load_case('mid')
    fc, att, labels, masks, top = to_dev(batch, dev)
    g = 3
    rep = lambda t: t[:2].repeat_interleave(g, dim=0).contiguous()  # noqa: E731   (2 images x 3 captions)
    fc_r, att_r = [rep(f) for f in fc], [rep(a) for a in att]
    crit = R.ReviewNetEnsembleCriterion(cfg)
    outs = []
    for dedup in (0, g):
        model = build(cfg, P, dev)
        model.dedup_seq_per_img = dedup
        lp, reason = model(fc_r, att_r, labels)
        crit(lp, labels[:, 1:], masks[:, 1:], reason, top, 1.0).backward()
        outs.append((lp.detach(), [r.detach() for r in reason], {k: p.grad.clone() for k, p in model.named_parameters()}))
    assert torch.equal(outs[0][0], outs[1][0])
    for a, b in zip(outs[0][1], outs[1][1]):
        assert torch.equal(a, b)
    for k in outs[0][2]:
        ref = outs[0][2][k]
        assert maxerr(outs[1][2][k], ref.cpu()) <= 1e-6 + 1e-4 * float(ref.abs().max()), k
    # the feeder's unique-image batch (FeatureFeeder.batch(expand=False)) composes with dedup: same outputs again
    model = build(cfg, P, dev)
    model.dedup_seq_per_img = g
    lp_u, reason_u = model([f[:2].contiguous() for f in fc], [a[:2].contiguous() for a in att], labels)
    assert torch.equal(lp_u, outs[0][0]) and all(torch.equal(a, b) for a, b in zip(reason_u, outs[0][1]))
    with pytest.raises(R._native.RfnError):
        model([f[:3] for f in fc], [a[:3] for a in att], labels)          # 3 images x 3 != 6 caption rows


def test_reuse_prefix_serves_the_baseline_sample_and_is_invalidated(dev):
    """model.reuse_prefix: the greedy baseline sample of the self-critical loop (train_rl.py:160-166) reuses the
    stage-I/II outputs of the multinomial sample on the same input tensors; any change of inputs or weights, a call
    that needs grad, or new tensor objects recompute."""
    import recurrent_fusion_network_amd as R
    import recurrent_fusion_network_amd.fusion_model as FM
    cfg, spec, P, batch, gold = load_case('mid')
    fc, att, labels, masks, top = to_dev(batch, dev)
    model = build(cfg, P, dev, train=True)
    with torch.no_grad():
        model.eval()
        want = model.sample(fc, att, {'sample_max': 1})
        model.train()
    calls = []
    orig = FM._PrefixFn.apply

    def counting(*a):
        calls.append(1)
        return orig(*a)

    FM._PrefixFn.apply = counting
    try:
        model.reuse_prefix = True
        seq, lp, lp_all, reason = model.sample(fc, att, {'sample_max': 0})          # with grad: computes, stores
        assert len(calls) == 1 and lp.requires_grad
        with torch.no_grad():
            model.eval()
            got = model.sample(fc, att, {'sample_max': 1})                          # baseline: served from the cache
            model.train()
        assert len(calls) == 1
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]) and torch.equal(got[2], want[2])
        lp.sum().backward()                                                         # the first graph is intact
        assert model.logit.weight.grad is not None
        seq2 = model.sample(fc, att, {'sample_max': 0})                             # needs grad: never served
        assert len(calls) == 2
        opt = R.FusedClampAdam(model, lr=1e-3)
        opt.step()                                                                  # weights moved -> stale
        with torch.no_grad():
            model.sample(fc, att, {'sample_max': 1})
        assert len(calls) == 3
        with torch.no_grad():
            model.sample(fc, att, {'sample_max': 1})                                # unchanged again: hit
            assert len(calls) == 3
            att[0].mul_(1.0)                                                        # in-place write bumps _version
            model.sample(fc, att, {'sample_max': 1})
            assert len(calls) == 4
            att2 = [a.clone() for a in att]                                         # same values, new objects
            model.sample(fc, att2, {'sample_max': 1})
            assert len(calls) == 5
        model.reuse_prefix = False
        with torch.no_grad():
            model.sample(fc, att2, {'sample_max': 1})
        assert len(calls) == 6
    finally:
        FM._PrefixFn.apply = orig


def test_inference_hooks(dev):
    """get_init_state / get_thought_vectors / one_time_step (misc/RecurrentFusionModel.py:283-350)."""
    from oracle import rfn_oracle as O
    cfg, spec, P, batch, gold = load_case('mid')
    model = build(cfg, P, dev)
    fc, att, labels, masks, top = to_dev(batch, dev)
    state = model.get_init_state(fc)
    hs, cs = O.init_state(cfg, P, batch[0])
    for i in range(len(hs)):
        assert maxerr(state[i][0][0], hs[i]) < 1e-4
    comb, reason, st = model.get_thought_vectors(fc, att, state)
    o_comb, o_reason, (oh, oc) = O.thought_vectors(cfg, P, batch[1], hs, cs)
    assert maxerr(comb, o_comb) < 1e-4 and maxerr(st[0][0], oh) < 1e-4
    comb0, reason0, st0 = model.get_thought_vectors(fc, att)              # state recomputed from fc_feats: same bits
    assert torch.equal(comb0, comb) and torch.equal(st0[1], st[1])
    # a state_list that is NOT what get_init_state returns is honoured (c != h, scaled h), like the reference
    hs2 = [h * 0.5 for h in hs]
    cs2 = [c * -0.25 for c in cs]
    state2 = [(h.unsqueeze(0).to(dev), c.unsqueeze(0).to(dev)) for h, c in zip(hs2, cs2)]
    comb2, reason2, st2_ = model.get_thought_vectors(fc, att, state2)
    o_comb2, o_reason2, (oh2_, oc2_) = O.thought_vectors(cfg, P, batch[1], hs2, cs2)
    assert maxerr(comb2, o_comb2) < 1e-4 and maxerr(st2_[1][0], oc2_) < 1e-4
    for a, b in zip(reason2, o_reason2):
        assert maxerr(a, b) < 1e-4
    assert maxerr(comb2, o_comb) > 1e-3
    ids = torch.zeros(fc[0].size(0), dtype=torch.long, device=dev)
    logit, st2 = model.one_time_step(ids, fc, comb, st)
    o_logit, oh2, oc2 = O.one_time_step(cfg, P, P['embed.weight'][ids.cpu()], o_comb, oh, oc)
    assert maxerr(logit, o_logit) < 1e-4 and maxerr(st2[1][0], oc2) < 1e-4
    # the reference's literal calling convention: the caller embeds the token itself (eval_utils.py:368)
    ids2 = torch.arange(1, fc[0].size(0) + 1, device=dev)
    with torch.no_grad():
        xt = model.embed(ids2)
    assert tuple(xt.shape) == (fc[0].size(0), cfg.input_encoding_size)
    logit_x, st_x = model.one_time_step(xt, fc, comb, st)
    logit_i, st_i = model.one_time_step(ids2, fc, comb, st)
    assert torch.equal(logit_x, logit_i) and torch.equal(st_x[0], st_i[0]) and torch.equal(st_x[1], st_i[1])
    with pytest.raises(Exception):
        model.one_time_step(xt[:, :-1], fc, comb, st)


def test_unchanged_trainer_route_torch_adam_and_clip_gradient(dev):
    """train.py:143-163 unchanged: zero_grad -> forward -> crit -> backward -> utils.clip_gradient -> optim.Adam.step.
    `.grad` must be populated for every parameter and the update must equal the fused optimizer's."""
    import recurrent_fusion_network_amd as R
    cfg, spec, P, batch, gold = load_case('mid')
    fc, att, labels, masks, top = to_dev(batch, dev)
    crit = R.ReviewNetEnsembleCriterion(cfg)
    m1 = build(cfg, P, dev, train=True)
    opt1 = torch.optim.Adam(m1.parameters(), lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5)
    m2 = build(cfg, P, dev, train=True)
    opt2 = R.FusedClampAdam(m2, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5, grad_clip=0.05)
    for step in range(2):
        opt1.zero_grad()
        lp, reason = m1(fc, att, labels)
        crit(lp, labels[:, 1:], masks[:, 1:], reason, top, 1.0).backward()
        assert all(p.grad is not None for p in m1.parameters())
        R.clip_gradient(opt1, 0.05)                       # small clip so the clamp really bites
        opt1.step()
        opt2.zero_grad()
        lp2, reason2 = m2(fc, att, labels)
        crit(lp2, labels[:, 1:], masks[:, 1:], reason2, top, 1.0).backward()
        opt2.step()
    p2 = dict(m2.named_parameters())
    for k, p in m1.named_parameters():
        assert maxerr(p, p2[k].detach().cpu()) < 3e-6, k
    # a second backward before zero_grad accumulates, like autograd
    opt1.zero_grad()
    for _ in range(2):
        lp, reason = m1(fc, att, labels)
        crit(lp, labels[:, 1:], masks[:, 1:], reason, top, 1.0).backward()
    g2 = m1.logit.weight.grad.clone()
    opt1.zero_grad()
    lp, reason = m1(fc, att, labels)
    crit(lp, labels[:, 1:], masks[:, 1:], reason, top, 1.0).backward()
    assert maxerr(g2, (2 * m1.logit.weight.grad).cpu()) < 1e-6 + 1e-5 * float(g2.abs().max())


def test_checkpoint_round_trip_and_models_setup(dev, tmp_path):
    """models.setup(opt) with start_from / load_model_id (models.py:26-36): a saved state_dict reloads to the same
    outputs."""
    import pickle
    import recurrent_fusion_network_amd as R
    cfg, spec, P, batch, gold = load_case('tiny1')
    fc, att, labels, masks, top = to_dev(batch, dev)
    model = build(cfg, P, dev)
    with torch.no_grad():
        ref, _ = model(fc, att, labels)
    torch.save(model.state_dict(), str(tmp_path / 'model_best.pth'))
    with open(str(tmp_path / 'infos_best.pkl'), 'wb') as f:
        pickle.dump({'iter': 0}, f)
    cfg.start_from, cfg.load_model_id = str(tmp_path), 'best'
    m2 = R.setup(cfg).to(dev).eval()
    with torch.no_grad():
        out, _ = m2(fc, att, labels)
    assert torch.equal(out, ref)


def test_scheduled_sampling(dev):
    """misc/RecurrentFusionModel.py:260-270: ss_prob -> 0 is teacher forcing; ss_prob = 1 feeds sampled tokens from
    the model's previous distribution (steps >= 1), so later log-probs change while step 0 cannot."""
    cfg, spec, P, batch, gold = load_case('mid')
    model = build(cfg, P, dev, train=True)
    fc, att, labels, masks, top = to_dev(batch, dev)
    base, _ = model(fc, att, labels)
    model.ss_prob = 1e-12
    almost, _ = model(fc, att, labels)
    assert torch.equal(base, almost)
    model.ss_prob = 1.0
    torch.manual_seed(3)
    ss, _ = model(fc, att, labels)
    assert tuple(ss.shape) == tuple(base.shape)
    assert torch.equal(ss[:, :1], base[:, :1])           # step 0 is fed BOS; from step 1 on the input is a sample
    assert not torch.equal(ss[:, 1:], base[:, 1:])
    ss.sum().backward()                                    # gradients flow through the sampled-token pass
    assert all(torch.isfinite(p.grad).all() for p in model.parameters())
    torch.manual_seed(3)
    again, _ = model(fc, att, labels)
    assert torch.equal(ss, again)


@pytest.mark.parametrize('name', ['mid', 'tinymax'])
def test_sampling_under_dropout_draws_from_the_differentiated_distribution(dev, name):
    """misc/RecurrentFusionModel.py:260-270, 623-631: in training mode the reference samples (scheduled sampling, and
    sample(sample_max=0)) from the outputs of the dropout-affected pass it differentiates.  Scheduled sampling here
    is ONE step-wise pass that is both sampled and differentiated (rfn_decoder_fwd_step), pinned against a batched
    teacher-forced replay of the tokens it fed; sample(sample_max=0) draws in a free-running pass and takes the gradient
    from a teacher-forced replay.  In both cases the two passes must apply the same dropout masks and produce the same
    log-probs bit for bit (the published recipe is drop_prob_lm 0.3 + scheduled sampling,
    train_recurrent_fusion_model.sh:25-26)."""
    cfg, spec, P, batch, gold = load_case(name)
    cfg.drop_prob_lm, cfg.drop_prob_reason, cfg.drop_prob_fusion = 0.3, 0.2, 0.1
    model = build(cfg, P, dev, train=True)
    fc, att, labels, masks, top = to_dev(batch, dev)
    from recurrent_fusion_network_amd.fusion_model import _fresh_seed
    model._trace_ss = True
    model.ss_prob = 1.0
    torch.manual_seed(3)
    lp, _ = model(fc, att, labels)
    fed = model._ss_ids                                          # the tokens the sampled pass ended up feeding
    assert torch.equal(fed[:, 0], labels[:, 0].to(dev)) and not torch.equal(fed[:, 1:], labels[:, 1:fed.size(1)].to(dev))
    # a teacher-forced (batched) pass on those tokens with the same dropout seed reproduces the sampled pass bit for bit
    torch.manual_seed(3)
    seed = _fresh_seed()
    with torch.no_grad():
        comb, h, c, _ = model._prefix(fc, att, True, seed)
        tf = model._decode_teacher_forced(fed, comb, h, c, True, seed)
    assert torch.equal(tf, lp.detach()), 'the sampled pass is not the pass a teacher-forced replay computes'
    eval_lp = build(cfg, P, dev)(fc, att, labels)[0]
    assert not torch.equal(lp[:, 0], eval_lp[:, 0])             # dropout really was active
    lp.sum().backward()
    assert all(torch.isfinite(p.grad).all() for p in model.parameters())
    # multinomial sample with grad (train_rl.py:160): one step-wise pass, pinned against a batched replay of its tokens
    model.ss_prob = 0.0
    torch.manual_seed(4)
    seq, seq_lp, lp_all, _ = model.sample(fc, att, {'sample_max': 0})
    fed = model._sample_ids
    assert lp_all.requires_grad and fed.size(1) == lp_all.size(1)
    torch.manual_seed(4)
    seed = _fresh_seed()
    with torch.no_grad():
        comb, h, c, _ = model._prefix(fc, att, True, seed)
        tf = model._decode_teacher_forced(fed, comb, h, c, True, seed)
    assert torch.equal(tf, lp_all.detach())
    picked = lp_all.detach()[:, :seq.size(1)].gather(2, fed[:, 1:seq.size(1) + 1].unsqueeze(2)).squeeze(2)
    assert torch.equal(seq_lp.detach(), picked)
    alive = seq > 0
    assert torch.equal(seq[alive], fed[:, 1:seq.size(1) + 1][alive])
    (seq_lp * alive).sum().backward()
    assert all(torch.isfinite(p.grad).all() for p in model.parameters())
    # eval mode: no dropout anywhere, teacher-forced and free-running passes still agree bit for bit
    model.eval()
    with torch.no_grad():
        g_seq, g_lp, g_all, _ = model.sample(fc, att, {'sample_max': 1})
        S = g_all.size(1)
        ids = torch.zeros(g_seq.size(0), S, dtype=torch.long, device=dev)
        ids[:, 1:] = g_all[:, :S - 1].argmax(2)
        comb, h, c, _ = model._prefix(fc, att, False, 0)
        tf = model._decode_teacher_forced(ids, comb, h, c, False, 0)
    assert torch.equal(tf, g_all)


def test_ensemble_decode_matches_oracle(dev):
    """eval_utils.py:268-290: mean of the members' logits, log_softmax, shared greedy token."""
    from oracle import rfn_oracle as O
    from recurrent_fusion_network_amd.ensemble import EnsembleDecoder
    cfg, spec, P, batch, gold = load_case('mid')
    Ps = [P, O.seeded_params(cfg, 41), O.seeded_params(cfg, 42)]
    models = [build(cfg, p, dev) for p in Ps]
    fc, att, labels, masks, top = to_dev(batch, dev)
    seq, seq_lp, lp_all = EnsembleDecoder(models).sample(fc, att)
    # oracle ensemble, free running
    B = batch[0][0].size(0)
    states = []
    for p in Ps:
        hs, cs = O.init_state(cfg, p, batch[0])
        comb, _, (h, c) = O.thought_vectors(cfg, p, batch[1], hs, cs)
        states.append([comb, h, c])
    it = torch.zeros(B, dtype=torch.long)
    o_seq, o_lp, unfinished = [], [], None
    for t in range(cfg.seq_length + 1):
        if t >= 1:
            val, it = torch.max(logprobs, 1)
            unfinished = (it > 0) if t == 1 else unfinished & (it > 0)
            if int(unfinished.sum()) == 0:
                break
            o_seq.append(it * unfinished.long())
            o_lp.append(val)
        logits = []
        for p, s_ in zip(Ps, states):
            lg, s_[1], s_[2] = O.one_time_step(cfg, p, p['embed.weight'][it], s_[0], s_[1], s_[2])
            logits.append(lg)
        logprobs = torch.log_softmax(sum(logits) / len(Ps), 1)
    assert torch.equal(seq.cpu(), torch.stack(o_seq, 1))
    assert maxerr(seq_lp, torch.stack(o_lp, 1)) < LOGP_TOL
    # members spread over ranks: the logit sum goes through one RCCL all-reduce per step (1-rank group here)
    import torch.distributed as dist
    if not dist.is_initialized():
        dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29533', rank=0, world_size=1,
                                device_id=dev)
        try:
            seq_pg = EnsembleDecoder(models, process_group=dist.group.WORLD).sample(fc, att)[0]
            assert torch.equal(seq_pg, seq)
        finally:
            dist.destroy_process_group()
    # a one-member "ensemble" is the plain greedy sample
    s1 = EnsembleDecoder(models[:1]).sample(fc, att)[0]
    with torch.no_grad():
        assert torch.equal(s1, models[0].sample(fc, att, {'sample_max': 1})[0])


def test_ensemble_beam_search_matches_oracle(dev):
    """eval_utils.eval_ensemble (eval_utils.py:387-720): the fusion model's beam search on the mean of the members'
    logits.  Three members against the oracle's member-list search; one member against the model's own sample_beam
    (which reference-generated goldens pin), bit for bit."""
    from oracle import rfn_oracle as O
    from recurrent_fusion_network_amd.ensemble import EnsembleDecoder
    cfg, spec, P, batch, gold = load_case('mid')
    Ps = [P, O.seeded_params(cfg, 41), O.seeded_params(cfg, 42)]
    models = [build(cfg, p, dev) for p in Ps]
    fc, att, labels, masks, top = to_dev(batch, dev)
    ens = EnsembleDecoder(models)
    seq, seq_lp, top_seq, top_prob = ens.sample_beam(fc, att, {'beam_size': 3})
    o_seq, o_lp, o_top_seq, o_top_prob, _, o_done = O.sample_beam(cfg, Ps, batch[0], batch[1], 3)
    assert torch.equal(seq.cpu(), o_seq)
    assert maxerr(seq_lp, o_lp) < LOGP_TOL
    for k in range(len(o_top_seq)):
        assert torch.equal(top_seq[k], o_top_seq[k]), k
        assert max(abs(a - b) for a, b in zip(top_prob[k], o_top_prob[k])) < 1e-3
        assert len(ens.done_beams[k]) == len(o_done[k])
    # the members disagree with each other: the ensemble's captions are not simply member 0's
    with torch.no_grad():
        own = [m.sample(fc, att, {'beam_size': 3})[0] for m in models]
    assert any(not torch.equal(o, seq) for o in own)
    # one member: the model's own search
    s1, l1, t1, p1 = EnsembleDecoder(models[:1]).sample_beam(fc, att, {'beam_size': 3})
    with torch.no_grad():
        m_seq, m_lp, m_top_seq, m_top_prob, _ = models[0].sample(fc, att, {'beam_size': 3})
    assert torch.equal(s1, m_seq) and torch.equal(l1, m_lp)
    assert all(torch.equal(a, b) for a, b in zip(t1, m_top_seq)) and list(p1) == list(m_top_prob)


def test_cpu_inputs_fail_loudly():
    import recurrent_fusion_network_amd as R
    cfg, spec, P, batch, gold = load_case('tiny0')
    model = R.RecurrentFusionModel(cfg)
    with pytest.raises(R._native.RfnError):
        model(batch[0], batch[1], batch[2])


def test_beam_search_on_long_captions(dev):
    """seq_length = 40 (the beam kernel keeps up to 64 steps of history per image): against the oracle's per-image search."""
    from oracle import rfn_oracle as O
    info = [dict(att_num=L, att_feat_size=D, fc_feat_size=F) for (L, D, F) in [(5, 24, 24), (7, 40, 32)]]
    cfg = O.make_cfg(info, vocab_size=50, rnn_size=16, input_encoding_size=16, att_hid_size=16, num_review_steps_0=3,
                     num_review_steps=3, top_words_count=20, seq_length=40)
    P = O.seeded_params(cfg, 21, scale=0.3)
    fc, att, labels, masks, top = O.synthetic_batch(cfg, 3, seed=77)
    want_seq, want_lp, want_top_seq, want_top_prob, _, _ = O.sample_beam(cfg, P, fc, att, 3)
    model = build(cfg, P, dev)
    seq, seq_lp, top_seq, top_prob, reason = model.sample([f.to(dev) for f in fc], [a.to(dev) for a in att], {'beam_size': 3})
    assert seq.shape == (3, 40)
    assert torch.equal(seq.cpu(), want_seq)
    assert maxerr(seq_lp, want_lp) < LOGP_TOL
    assert int((want_seq != 0).sum(1).max()) > 32          # the case does run past the old 32-step limit
    for k in range(3):
        assert torch.equal(top_seq[k], want_top_seq[k])
        assert np.allclose(np.array(top_prob[k]), np.array(want_top_prob[k]), atol=1e-3)


@pytest.mark.parametrize('beam', [20, 32])
def test_beam_search_with_wide_beams(dev, beam):
    """beam_size above 16 (the device bookkeeping serves up to 32 beams per image): against the oracle's per-image search."""
    from oracle import rfn_oracle as O
    info = [dict(att_num=L, att_feat_size=D, fc_feat_size=F) for (L, D, F) in [(5, 24, 24), (7, 40, 32)]]
    cfg = O.make_cfg(info, vocab_size=50, rnn_size=16, input_encoding_size=16, att_hid_size=16, num_review_steps_0=3,
                     num_review_steps=3, top_words_count=20, seq_length=6)
    P = O.seeded_params(cfg, 5, scale=0.4)
    fc, att, labels, masks, top = O.synthetic_batch(cfg, 2, seed=78)
    want_seq, want_lp, want_top_seq, want_top_prob, _, _ = O.sample_beam(cfg, P, fc, att, beam)
    model = build(cfg, P, dev)
    seq, seq_lp, top_seq, top_prob, reason = model.sample([f.to(dev) for f in fc], [a.to(dev) for a in att], {'beam_size': beam})
    assert torch.equal(seq.cpu(), want_seq)
    assert maxerr(seq_lp, want_lp) < LOGP_TOL
    for k in range(2):
        assert torch.equal(top_seq[k], want_top_seq[k])
        assert np.allclose(np.array(top_prob[k]), np.array(want_top_prob[k]), atol=1e-3)
