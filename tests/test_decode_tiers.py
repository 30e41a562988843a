"""Decode tiers (BASELINE config 5 and the evaluation loop): beam search with the config's beam size, the eval-loop
body (XE loss on the seq_per_img-replicated caption batch, sample on rows arange(n) * seq_per_img, sentence score
sum(seqLogprobs * (seq > 0)); eval_utils.py:149-151,159-208) and the self-critical sample path (train_rl.py:160-191),
against fixtures captured from the reference by oracle/make_golden.py (generate_decode).
  evalmid: heterogeneous mid shape, 2 images x 5 captions, beam 5        (CPU oracle + GPU)
  c5     : C3-shaped (M=4, L=196, D=2048), 3 images x 2 captions, beam 5  (GPU; the CPU oracle pin is marked slow)
plus properties at the stated size (B=128 images, beam 5, M=4): batch independence and determinism."""
import numpy as np
import pytest
import torch

from conftest import load_case


def maxerr(a, b):
    return float((torch.as_tensor(a).detach().double().cpu() - torch.as_tensor(b).double()).abs().max())


def _caption_rows(spec, batch):
    spi = spec['decode']['spi']
    fc, att, labels, masks, top = batch
    rep = lambda t: t[::spi].repeat_interleave(spi, 0).contiguous()  # noqa: E731
    return [rep(f) for f in fc], [rep(a) for a in att], labels, masks, top


# ---------------------------------------------------------------------------------------------------------------
# CPU: the oracle against the reference's captured outputs
# ---------------------------------------------------------------------------------------------------------------
def test_oracle_eval_loop_and_beam5_and_rl_evalmid():
    from oracle import rfn_oracle as O
    cfg, spec, P, batch, gold = load_case('evalmid')
    spi, beam = int(gold['seq_per_img']), int(gold['beam_size'])
    fc, att, labels, masks, top = _caption_rows(spec, batch)
    loss, seq, seq_lp, sent = O.eval_step(cfg, P, fc, att, labels, masks, top, spi, 1.0, 1)
    assert abs(float(loss) - float(gold['eval_xe_loss'])) < 1e-4
    assert torch.equal(seq, torch.from_numpy(gold['eval_greedy_seq']))
    assert maxerr(seq_lp, gold['eval_greedy_seq_logprobs']) < 2e-5 and maxerr(sent, gold['eval_greedy_sentence']) < 1e-4
    _, bseq, blp, bsent = O.eval_step(cfg, P, fc, att, labels, masks, top, spi, 1.0, beam)
    assert torch.equal(bseq, torch.from_numpy(gold['beam_seq']))
    assert maxerr(blp, gold['beam_seq_logprobs']) < 2e-5 and maxerr(bsent, gold['beam_sentence']) < 1e-4
    rows = torch.arange(len(fc[0]) // spi) * spi
    fc_u, att_u, top_u = [f[rows] for f in fc], [a[rows] for a in att], top[rows]
    with torch.no_grad():
        rs, rlp, rall, rreason = O.sample_greedy(cfg, P, fc_u, att_u, force_ids=torch.from_numpy(gold['rl_raw_ids']))
    assert torch.equal(rs, torch.from_numpy(gold['rl_seq']))
    rl = O.rl_criterion(cfg, rlp, rs, torch.from_numpy(gold['rl_reward']), rall, 0.01, rreason, top_u, 1.0)
    assert abs(float(rl) - float(gold['rl_loss'])) < 1e-4


def test_eval_shim_row_selection_and_errors():
    import recurrent_fusion_network_amd as R
    assert R.unique_image_rows(15, 5).tolist() == [0, 5, 10]
    with pytest.raises(ValueError):
        R.unique_image_rows(14, 5)


# ---------------------------------------------------------------------------------------------------------------
# GPU: the HIP path against the same fixtures
# ---------------------------------------------------------------------------------------------------------------
def _build(cfg, P, dev, train=False, gemm='exact'):
    import recurrent_fusion_network_amd as R
    model = R.RecurrentFusionModel(cfg)
    model.load_state_dict(P)
    if gemm == 'bf16x3':     # the hoisted projections / their weight gradients on the bf16 matrix cores, at any size
        import recurrent_fusion_network_amd._native as N
        model.gemm_flags |= N.GEMM_OPT_BF16X3 | N.GEMM_OPT_BF16X3_ANY_SIZE
    return model.to(dev).train(train)


@pytest.mark.gpu
@pytest.mark.parametrize('name,gemm', [('evalmid', 'exact'), ('c5', 'exact'), ('c5', 'bf16x3')])
def test_eval_loop_greedy_and_beam5(dev, name, gemm):
    import recurrent_fusion_network_amd as R
    cfg, spec, P, batch, gold = load_case(name)
    spi, beam = int(gold['seq_per_img']), int(gold['beam_size'])
    fc, att, labels, masks, top = [[x.to(dev) for x in t] if isinstance(t, list) else t.to(dev)
                                   for t in _caption_rows(spec, batch)]
    model = _build(cfg, P, dev, gemm=gemm)
    crit = R.ReviewNetEnsembleCriterion(cfg)
    out = R.eval_step(model, crit, fc, att, labels, masks, top, spi, 1.0, beam_size=1)
    assert abs(float(out['loss']) - float(gold['eval_xe_loss'])) < 1e-4 * max(1.0, abs(float(gold['eval_xe_loss'])))
    assert torch.equal(out['seq'].cpu(), torch.from_numpy(gold['eval_greedy_seq']))      # greedy ids: bit-exact
    assert maxerr(out['seqLogprobs'], gold['eval_greedy_seq_logprobs']) < 1e-3
    assert maxerr(out['log_probs_sentence'], gold['eval_greedy_sentence']) < 1e-3
    assert len(out['sample']) == 4
    outb = R.eval_step(model, crit, fc, att, labels, masks, top, spi, 1.0, beam_size=beam)
    assert len(outb['sample']) == 5                                                       # eval_utils.py:198-200
    assert torch.equal(outb['seq'].cpu(), torch.from_numpy(gold['beam_seq']))
    assert maxerr(outb['seqLogprobs'], gold['beam_seq_logprobs']) < 1e-3
    assert maxerr(outb['log_probs_sentence'], gold['beam_sentence']) < 1e-3
    top_seq, top_prob = outb['sample'][2], outb['sample'][3]
    for k in range(gold['beam_seq'].shape[0]):
        assert torch.equal(top_seq[k], torch.from_numpy(gold['beam_top_seq_%d' % k]))
        assert np.allclose(np.array(top_prob[k]), gold['beam_top_prob_%d' % k], atol=1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize('name,gemm', [('evalmid', 'exact'), ('c5', 'exact'), ('c5', 'bf16x3')])
def test_self_critical_sample_path(dev, name, gemm):
    """train_rl.py:160-191 at the tier's shape: multinomial sample with grad (ids replayed from the reference's draw),
    greedy baseline (get_rewards.py:119-126), reward criterion, backward; gradient norms and slices vs the reference."""
    import recurrent_fusion_network_amd as R
    cfg, spec, P, batch, gold = load_case(name)
    spi = int(gold['seq_per_img'])
    fc, att, labels, masks, top = _caption_rows(spec, batch)
    rows = torch.arange(len(fc[0]) // spi) * spi
    fc_u, att_u, top_u = [f[rows].to(dev) for f in fc], [a[rows].to(dev) for a in att], top[rows].to(dev)
    model = _build(cfg, P, dev, gemm=gemm)
    seq, seq_lp, lp_all, reason = model.sample(fc_u, att_u, {'sample_max': 0, 'force_ids': torch.from_numpy(gold['rl_raw_ids'])})
    assert torch.equal(seq.cpu(), torch.from_numpy(gold['rl_seq']))
    assert maxerr(seq_lp, gold['rl_seq_logprobs']) < 1e-3
    with torch.no_grad():
        base = model.sample(fc_u, att_u, {})[0]                                           # the greedy baseline
    assert torch.equal(base.cpu(), torch.from_numpy(gold['eval_greedy_seq']))
    crit = R.ReviewNetRewardCriterion(cfg)
    loss = crit(seq_lp, seq, torch.from_numpy(gold['rl_reward']).to(dev), lp_all, 0.01, reason, top_u, 1.0, None, cfg)
    assert abs(float(loss.detach()) - float(gold['rl_loss'])) < 1e-4 * max(1.0, abs(float(gold['rl_loss'])))
    loss.backward()
    for k, p in model.named_parameters():
        gn = float(gold['rl_gradnorm/' + k])
        assert abs(float(p.grad.double().norm()) - gn) <= 1e-5 + 3e-3 * gn, k
        g = p.grad.detach().reshape(-1).cpu()
        stride = max(1, g.numel() // 16)
        sl = gold['rl_gradslice/' + k]
        assert maxerr(g[::stride][:16], sl) <= 1e-5 + 3e-3 * max(float(np.abs(sl).max()), gn / max(1.0, g.numel() ** 0.5)), k


@pytest.mark.gpu
def test_config5_size_beam5_and_rl_sample_properties(dev):
    """BASELINE configs[4] at its stated size (M=4, L=196, D=2048, B=128 images, beam 5): results are independent of
    the batch an image sits in (rows 3..5 alone == the same rows inside the 128-image batch), deterministic from call
    to call, the best beam scores are sorted, and the multinomial sample's differentiable log-probs equal the
    free-running ones."""
    import bench as BN
    import recurrent_fusion_network_amd as R
    w = dict(BN.WORKLOADS['c5'])
    cfg = BN.make_cfg(w)
    model = R.RecurrentFusionModel(cfg).to(dev)
    BN.seeded_weights_(model, 100)
    model.eval()
    fc, att, labels, masks, top = BN.synthetic_inputs(cfg, w['B'], 100, dev)
    with torch.no_grad():
        a = model.sample(fc, att, {'beam_size': 5})
        b = model.sample(fc, att, {'beam_size': 5})
        part = model.sample([f[3:6] for f in fc], [x[3:6] for x in att], {'beam_size': 5})
        g = model.sample(fc, att, {'sample_max': 1})
        gp = model.sample([f[3:6] for f in fc], [x[3:6] for x in att], {'sample_max': 1})
    assert tuple(a[0].shape) == (w['B'], cfg.seq_length)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])                    # deterministic
    assert torch.equal(part[0], a[0][3:6]) and maxerr(part[1], a[1][3:6].cpu()) < 1e-5
    for k in (0, 64, 127):
        p = np.array(a[3][k])
        assert len(p) >= 1 and np.all(np.diff(p) <= 1e-6)
        assert torch.equal(a[2][k][0], a[0][k].cpu())                             # returned seq = best done beam
    # greedy ids do not depend on the batch an image sits in (log-probs to rounding: split-K choices follow the row count)
    assert torch.equal(gp[0], g[0][3:6]) and maxerr(gp[1], g[1][3:6].cpu()) < 1e-5
    # beam search never scores below the greedy sentence when the greedy sentence ended inside the beam
    model.train()
    model._trace_ss = True
    torch.manual_seed(5)
    seq, seq_lp, lp_all, _ = model.sample(fc, att, {'sample_max': 0})
    fed = model._sample_ids
    assert lp_all.requires_grad and tuple(fed.shape) == (w['B'], lp_all.size(1))
    with torch.no_grad():     # the sampled pass equals a batched teacher-forced replay of the tokens it fed (dropout is 0)
        comb, h, c, _ = model._prefix(fc, att, True, 0)
        tf = model._decode_teacher_forced(fed, comb, h, c, True, 0)
    assert torch.equal(tf, lp_all.detach())
