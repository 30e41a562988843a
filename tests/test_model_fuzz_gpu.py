"""Randomised whole-path parity: two dozen seeded random model configurations (encoder count, every width, step counts,
maxout, label smoothing, ragged captions, batch size) -- HIP path vs the CPU oracle on forward log-probs, the XE loss,
EVERY parameter gradient and the greedy token ids (bit-exact).  The oracle itself is pinned by the reference's golden
vectors on seven fixed tiers (tests/test_oracle_golden.py); this sweeps the shape space between them."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _random_case(seed, x3=False):
    """x3: shapes the bf16-plane GEMM accepts (att_hid_size a multiple of its 256-wide output groups, feature widths
    multiples of 4), everything else as random as before -- so RFN_GEMM_OPT_BF16X3 | _ANY_SIZE really takes it."""
    from oracle import rfn_oracle as O
    rng = np.random.default_rng(seed + (5000 if x3 else 0))
    M = int(rng.integers(1, 5))
    info = [dict(att_num=int(rng.integers(1, 41)), att_feat_size=int(rng.integers(1, 18)) * 4 if x3 else int(rng.integers(4, 71)),
                 fc_feat_size=int(rng.integers(4, 51))) for _ in range(M)]
    extra = dict(review_maxout=int(rng.integers(0, 2)), maxout=int(rng.integers(0, 2)),
                 use_label_smoothing=int(rng.integers(0, 2)), label_smoothing_epsilon=0.1)
    cfg = O.make_cfg(info, vocab_size=int(rng.integers(20, 121)), rnn_size=int(rng.integers(8, 41)),
                     input_encoding_size=int(rng.integers(8, 41)), att_hid_size=256 if x3 else int(rng.integers(8, 41)),
                     num_review_steps_0=int(rng.integers(1, 7)), num_review_steps=int(rng.integers(1, 7)),
                     top_words_count=int(rng.integers(5, 31)), seq_length=int(rng.integers(2, 9)), **extra)
    B = int(rng.integers(1, 7))
    P = O.seeded_params(cfg, 100 + seed, scale=0.08)
    fc, att, labels, masks, top = O.synthetic_batch(cfg, B, seed=200 + seed)
    S = cfg.seq_length
    n = rng.integers(1, S + 1, size=B)                     # ragged captions: row b keeps n_b words
    for b in range(B):
        labels[b, 1 + n[b]:] = 0
        masks[b, n[b] + 2:] = 0
    return O, cfg, P, (fc, att, labels, masks, top)


@pytest.mark.parametrize('gemm', ['exact', 'bf16x3'])
@pytest.mark.parametrize('seed', range(24))
def test_random_configuration_matches_oracle(dev, seed, gemm):
    """gemm = bf16x3: the same 24 seeds on shapes the plane GEMM takes, with RFN_GEMM_OPT_BF16X3 | _ANY_SIZE: same bars
    against the oracle as the exact path, greedy ids included."""
    import recurrent_fusion_network_amd as R
    O, cfg, P, (fc, att, labels, masks, top) = _random_case(seed, x3=(gemm == 'bf16x3'))
    model = R.RecurrentFusionModel(cfg)
    model.load_state_dict(P)
    model = model.to(dev).eval()
    if gemm == 'bf16x3':
        import recurrent_fusion_network_amd._native as N
        model.gemm_flags |= N.GEMM_OPT_BF16X3 | N.GEMM_OPT_BF16X3_ANY_SIZE
    d = lambda ts: [t.to(dev) for t in ts]  # noqa: E731
    lp, reason = model(d(fc), d(att), labels.to(dev))
    crit = R.ReviewNetEnsembleCriterion(cfg)
    loss = crit(lp, labels.to(dev)[:, 1:], masks.to(dev)[:, 1:], reason, top.to(dev), 1.0)
    loss.backward()
    o_lp, o_reason = O.forward(cfg, P, fc, att, labels)
    o_loss, o_grads = O.train_step_loss_and_grads(cfg, P, fc, att, labels, masks, top, 1.0)
    assert lp.shape == o_lp.shape, (tuple(lp.shape), tuple(o_lp.shape))
    assert float((lp.detach().cpu() - o_lp).abs().max()) < 1e-3
    for a, b in zip(reason, o_reason):
        assert float((a.detach().cpu() - b).abs().max()) < 1e-4
    assert abs(float(loss.detach()) - float(o_loss)) < 1e-4 * max(1.0, abs(float(o_loss)))
    named = dict(model.named_parameters())
    assert set(named) == set(o_grads)
    for k, g in o_grads.items():
        err = float((named[k].grad.cpu() - g).abs().max())
        assert err <= 1e-6 + 1e-3 * float(g.abs().max()), (k, err, float(g.abs().max()))
    with torch.no_grad():
        seq, seq_lp, lp_all, _ = model.sample(d(fc), d(att), {'sample_max': 1})
    o_seq = O.sample_greedy(cfg, P, fc, att)
    assert torch.equal(seq.cpu(), o_seq[0])
    assert seq_lp.shape == o_seq[1].shape
    if seq_lp.numel():                     # every row may emit END at once: an empty (B, 0) sample, as the reference
        assert float((seq_lp.cpu() - o_seq[1]).abs().max()) < 1e-3


@pytest.mark.parametrize('gemm', ['exact', 'bf16x3'])
@pytest.mark.parametrize('M,B', [(4, 24), (4, 23), (3, 33), (2, 50), (4, 31)])
def test_mid_size_batches_of_uniform_encoders(dev, M, B, gemm):
    """The batch sizes of data-parallel shards: uniform encoders with maps too big for the small-map rule (L * D > 32768),
    M * B on both sides of the threshold at which the stage-I attention backward of all encoders runs as ONE fused launch
    (csrc/rfn_path.hip attn_bwd_grouped: 96 blocks; with bf16x3 that launch also writes dP1 as bf16 planes).  Every
    gradient against the oracle, greedy ids bit-exact."""
    import recurrent_fusion_network_amd as R
    from oracle import rfn_oracle as O
    L, D = 70, 512
    info = [dict(att_num=L, att_feat_size=D, fc_feat_size=40) for _ in range(M)]
    cfg = O.make_cfg(info, vocab_size=90, rnn_size=32, input_encoding_size=24, att_hid_size=256, num_review_steps_0=2,
                     num_review_steps=2, top_words_count=12, seq_length=5)
    P = O.seeded_params(cfg, 900 + B, scale=0.08)
    fc, att, labels, masks, top = O.synthetic_batch(cfg, B, seed=901 + B)
    model = R.RecurrentFusionModel(cfg)
    model.load_state_dict(P)
    model = model.to(dev).eval()
    if gemm == 'bf16x3':
        import recurrent_fusion_network_amd._native as N
        model.gemm_flags |= N.GEMM_OPT_BF16X3 | N.GEMM_OPT_BF16X3_ANY_SIZE
    d = lambda ts: [t.to(dev) for t in ts]  # noqa: E731
    lp, reason = model(d(fc), d(att), labels.to(dev))
    loss = R.ReviewNetEnsembleCriterion(cfg)(lp, labels.to(dev)[:, 1:], masks.to(dev)[:, 1:], reason, top.to(dev), 1.0)
    loss.backward()
    o_lp, _ = O.forward(cfg, P, fc, att, labels)
    o_loss, o_grads = O.train_step_loss_and_grads(cfg, P, fc, att, labels, masks, top, 1.0)
    assert float((lp.detach().cpu() - o_lp).abs().max()) < 1e-3
    assert abs(float(loss.detach()) - float(o_loss)) < 1e-4 * max(1.0, abs(float(o_loss)))
    named = dict(model.named_parameters())
    for k, g in o_grads.items():
        err = float((named[k].grad.cpu() - g).abs().max())
        assert err <= 1e-6 + 1e-3 * float(g.abs().max()), (k, err, float(g.abs().max()))
    with torch.no_grad():
        seq = model.sample(d(fc), d(att), {'sample_max': 1})[0]
    assert torch.equal(seq.cpu(), O.sample_greedy(cfg, P, fc, att)[0])
