#!/usr/bin/env python3
"""Generate golden vectors by RUNNING THE REFERENCE in the build container.

Usage (build container only; /root/reference does not exist on the GPU box):
    python oracle/make_golden.py [--only tiny0,tiny1,mid,c2,c3,c5,evalmid,tiny0_drop,tinymax_drop,mid_drop]

Imports the reference's RecurrentFusionModel / criteria from /root/reference, loads the
documented seeded weight stream (oracle.rfn_oracle.seeded_params), runs forward / greedy sample /
beam / XE + RL criteria / backward / one clamp+Adam step, and stores ONLY tensors (inputs are
regenerated from seeds; digests of weights and inputs are stored so a drifting RNG stream is
detected) under tests/golden/*.npz.  No reference source, bytecode or pickled module is written.

It also asserts that the oracle restatement agrees with the reference on every stored quantity,
so a fixture can only be produced from a validated oracle.
"""
import argparse
import os
import sys
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.environ.get('RFN_REFERENCE', '/root/reference')

from oracle import rfn_oracle as O  # noqa: E402

CONFIGS = {
    # name: (feat specs (L, D, fc), R/A/E, V, K, T1, T2, B, seq_length, seed, ragged max words)
    'tiny0': dict(feats=[(5, 24, 24), (7, 40, 32)], R=16, V=50, K=20, T1=3, T2=3, B=3, S=5, seed=0, max_words=3),
    'tiny1': dict(feats=[(5, 24, 24), (7, 40, 32)], R=16, V=50, K=20, T1=3, T2=3, B=3, S=5, seed=1, max_words=5),
    # maxout variants of the stage-II and decoder cells (opts.py:180-185); fusion_maxout is set too and must be ignored
    'tinymax': dict(feats=[(5, 24, 24), (7, 40, 32)], R=16, V=50, K=20, T1=3, T2=3, B=3, S=5, seed=5, max_words=5,
                    extra=dict(review_maxout=1, maxout=1, fusion_maxout=1)),
    # R != A != E, every width off the 4-float vector paths, T1 != T2, heterogeneous L / D / fc
    'odd': dict(feats=[(9, 22, 13), (4, 35, 35), (11, 17, 29)], R=26, A=15, E=19, V=61, K=13, T1=3, T2=4, B=5, S=5,
                seed=6, max_words=4),
    'mid': dict(feats=[(196, 96, 96), (64, 80, 128), (49, 72, 72)], R=64, V=300, K=50, T1=8, T2=8, B=6, S=16,
                seed=2, max_words=16),
    'c2': dict(feats=[(49, 512, 512)] * 2, R=512, V=9487, K=1000, T1=8, T2=8, B=8, S=16, seed=3, max_words=16),
    'c3': dict(feats=[(196, 2048, 2048)] * 4, R=512, V=9487, K=1000, T1=8, T2=8, B=2, S=16, seed=4, max_words=16),
    # decode tiers (BASELINE config 5 and the eval loop, generate_decode): B caption rows = B / spi images, each image's
    # features repeated spi times in a row as the loader does; beam search with the config's beam size, RL sample
    # replay on the image rows.  'c5' is C3-shaped (M=4, L=196, D=2048), 'evalmid' the heterogeneous mid shape.
    'c5': dict(feats=[(196, 2048, 2048)] * 4, R=512, V=9487, K=1000, T1=8, T2=8, B=6, S=16, seed=8, max_words=12,
               decode=dict(spi=2, beam=5)),
    'evalmid': dict(feats=[(196, 96, 96), (64, 80, 128), (49, 72, 72)], R=64, V=300, K=50, T1=8, T2=8, B=10, S=16,
                    seed=9, max_words=9, decode=dict(spi=5, beam=5)),
}


def cfg_of(spec):
    info = [dict(att_num=L, att_feat_size=D, fc_feat_size=F) for (L, D, F) in spec['feats']]
    return O.make_cfg(info, vocab_size=spec['V'], rnn_size=spec['R'], input_encoding_size=spec.get('E', spec['R']),
                      att_hid_size=spec.get('A', spec['R']), num_review_steps_0=spec['T1'], num_review_steps=spec['T2'],
                      top_words_count=spec['K'], seq_length=spec['S'], **spec.get('extra', {}))


def batch_of(cfg, spec):
    fc, att, labels, masks, top = O.synthetic_batch(cfg, spec['B'], seed=1000 + spec['seed'])
    if spec['max_words'] < spec['S']:
        # ragged captions: row b keeps its first n_b words, n_b in [1, max_words]; exercises masks
        # and the all-zero-column early break (misc/RecurrentFusionModel.py:274)
        rng = np.random.default_rng(2000 + spec['seed'])
        n = rng.integers(1, spec['max_words'] + 1, size=spec['B'])
        for b in range(spec['B']):
            labels[b, 1 + n[b]:] = 0
            masks[b, n[b] + 2:] = 0
    return fc, att, labels, masks, top


def caption_rows(cfg, spec):
    """Decode tiers: the loader's batch layout (dataloader.py:247-252) -- row r holds the features of image
    r // spi, its own caption, mask and top words."""
    spi = spec['decode']['spi']
    fc, att, labels, masks, top = batch_of(cfg, spec)
    rep = lambda t: t[::spi].repeat_interleave(spi, 0).contiguous()  # noqa: E731
    return [rep(f) for f in fc], [rep(a) for a in att], labels, masks, top


def digest(tensors):
    acc = 0.0
    for t in tensors:
        a = t.detach().double().reshape(-1)
        w = torch.arange(1, a.numel() + 1, dtype=torch.float64) % 97 + 1
        acc += float((a * w).sum())
    return np.float64(acc)


def grad_summary(name, g):
    g = g.detach().reshape(-1)
    stride = max(1, g.numel() // 16)
    return np.float64(g.double().norm()), g[::stride][:16].numpy().copy()


def load_reference():
    sys.path.insert(0, REF)
    sys.argv = ['make_golden']
    warnings.filterwarnings('ignore')
    from misc.RecurrentFusionModel import RecurrentFusionModel  # noqa
    import misc.utils as ref_utils  # noqa
    return RecurrentFusionModel, ref_utils


class LegacyIndexing:
    """PyTorch-0.3.1 indexing semantics the reference's beam search relies on
    (misc/RecurrentFusionModel.py:473-478, 513): int-indexing a 1-D tensor or `[0]` on a 0-dim
    tensor yields a Python number (a COPY, not an aliasing view)."""

    def __enter__(self):
        self.orig = torch.Tensor.__getitem__
        orig = self.orig

        def getitem(t, idx):
            if isinstance(idx, int) and not isinstance(idx, bool):
                if t.dim() == 0 and idx == 0:
                    return t.item()
                if t.dim() == 1:
                    return orig(t, idx).item()
            return orig(t, idx)

        torch.Tensor.__getitem__ = getitem
        return self

    def __exit__(self, *a):
        torch.Tensor.__getitem__ = self.orig


def rl_section(model, ref_utils, cfg, spec, P, fc, att, top, out):
    """train_rl.py:160-191 on the reference: multinomial sample with grad, reward criterion, backward; the drawn ids
    are replayed through the oracle (asserted equal) and stored so the HIP path can replay them too."""
    B = fc[0].size(0)
    torch.manual_seed(77 + spec['seed'])
    model.zero_grad()
    s_seq, s_lp, s_all, s_rp = model.sample(fc, att, {'sample_max': 0, 'beam_size': 1, 'temperature': 1.0})
    T = s_seq.size(1)
    rng = np.random.default_rng(3000 + spec['seed'])
    reward = torch.from_numpy(np.repeat(rng.standard_normal((B, 1)).astype(np.float32), T, 1))
    rl_crit = ref_utils.ReviewNetRewardCriterion(cfg)
    rl_loss = rl_crit(s_lp, s_seq.data, reward, s_all, 0.01, s_rp, top, 1.0, None, cfg)
    rl_loss.backward()
    # replay the drawn ids through the oracle.  The reference masks ids of finished rows to 0 in
    # `seq` but feeds the UNMASKED draw to embed (:637,:647); finished rows never influence
    # other rows or unmasked loss terms, but logprobs_all of finished rows does enter the entropy
    # term only where mask_0 > 0, so replaying the masked ids is loss-equivalent only if no row
    # finishes early -- store the raw draws instead.
    torch.manual_seed(77 + spec['seed'])
    raw = []
    orig_multinomial = torch.multinomial

    def spy(*a, **k):
        r = orig_multinomial(*a, **k)
        raw.append(r.view(-1).clone())
        return r

    torch.multinomial = spy
    try:
        with torch.no_grad():
            model.sample(fc, att, {'sample_max': 0, 'beam_size': 1, 'temperature': 1.0})
    finally:
        torch.multinomial = orig_multinomial
    raw_ids = torch.stack(raw, 1)[:, :max(T, 1)]
    full_ids = torch.zeros(B, cfg.seq_length, dtype=torch.long)
    full_ids[:, :raw_ids.size(1)] = raw_ids
    Pg = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
    r_seq, r_lp, r_all, r_rp = O.sample_greedy(cfg, Pg, fc, att, force_ids=full_ids)
    assert torch.equal(r_seq, s_seq), 'RL replay ids differ'
    close(r_lp, s_lp, 2e-5, 'RL seqLogprobs')
    o_rl = O.rl_criterion(cfg, r_lp, r_seq, reward, r_all, 0.01, r_rp, top, 1.0)
    close(o_rl, rl_loss, 1e-4, 'RL loss')
    o_rl.backward()
    out['rl_raw_ids'] = full_ids.numpy()
    out['rl_seq'] = s_seq.numpy()
    out['rl_seq_logprobs'] = s_lp.detach().numpy()
    out['rl_reward'] = reward.numpy()
    out['rl_loss'] = np.float64(rl_loss.item())
    for k, p in model.named_parameters():
        close(Pg[k].grad if Pg[k].grad is not None else torch.zeros_like(Pg[k]), p.grad,
              2e-5 + 1e-4 * float(p.grad.abs().max()), 'RL grad ' + k)
        n, s = grad_summary(k, p.grad)
        out['rl_gradnorm/' + k] = n
        out['rl_gradslice/' + k] = s
    model.zero_grad()



def close(a, b, tol, what):
    err = float((a.detach().double() - b.detach().double()).abs().max())
    assert err <= tol, '%s: oracle vs reference max|diff| = %g > %g' % (what, err, tol)
    return err


def generate(name, RefModel, ref_utils, outdir):
    spec = CONFIGS[name]
    cfg = cfg_of(spec)
    full = name.startswith('tiny') or name == 'odd'
    torch.manual_seed(0)
    model = RefModel(cfg)
    sd = model.state_dict()
    shapes = O.param_shapes(cfg)
    assert list(sorted(sd.keys())) == list(sorted(shapes.keys())), 'state_dict schema mismatch'
    for k in sd:
        assert tuple(sd[k].shape) == shapes[k], (k, tuple(sd[k].shape), shapes[k])
    P = O.seeded_params(cfg, spec['seed'])
    model.load_state_dict(P)
    model.eval()
    fc, att, labels, masks, top = batch_of(cfg, spec)
    out = dict(weights_digest=digest([P[k] for k in sorted(P)]), inputs_digest=digest(fc + att),
               labels=labels.numpy(), masks=masks.numpy(), top_words=top.numpy())

    # ---- XE forward + criterion + backward + one optimiser step (train.py:143-163) ------------
    crit = ref_utils.ReviewNetEnsembleCriterion(cfg)
    optim = torch.optim.Adam(model.parameters(), lr=5e-4, betas=(0.9, 0.999), weight_decay=1e-5)
    optim.zero_grad()
    log_prob, top_pred = model(fc, att, labels)
    loss = crit(log_prob, labels[:, 1:], masks[:, 1:], top_pred, top, 1.0)
    loss.backward()
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    o_lp, o_rp = O.forward(cfg, P, fc, att, labels)
    e1 = close(o_lp, log_prob, 2e-5, 'log_prob')
    for a, b in zip(o_rp, top_pred):
        close(a, b, 2e-5, 'reason_pred')
    o_loss, o_grads = O.train_step_loss_and_grads(cfg, P, fc, att, labels, masks, top, 1.0)
    close(o_loss, loss, 1e-4, 'xe loss')
    for k in grads:
        close(o_grads[k], grads[k], 2e-5 + 1e-4 * float(grads[k].abs().max()), 'grad ' + k)
    ref_utils.clip_gradient(optim, 1.0)
    optim.step()
    stepped = {k: v.detach().clone() for k, v in model.state_dict().items()}
    o_P = {k: v.clone() for k, v in P.items()}
    O.clip_and_adam(o_P, o_grads, {}, lr=5e-4, weight_decay=1e-5, grad_clip=1.0)
    for k in stepped:
        # The first Adam step is lr*g/(|g|+eps): wherever |g| is not >> eps = 1e-8 it amplifies
        # rounding noise up to the full lr, so two correct implementations legitimately differ
        # there (att_h_2_out.bias is the extreme case: softmax shift invariance makes its exact
        # gradient 0).  Compare only elements with |g| > 1e-5 (noise/|g| < 1e-3).
        sel = grads[k].abs() > 1e-5
        if bool(sel.any()):
            close(o_P[k][sel], stepped[k][sel], 2e-6, 'adam ' + k)
    model.load_state_dict(P)

    out['log_prob_shape'] = np.array(log_prob.shape)
    out['xe_loss'] = np.float64(loss.item())
    for j, r in enumerate(top_pred):
        out['reason_pred_%d' % j] = r.detach().numpy() if (full or name == 'mid') else r.detach().numpy()[:, :32]
        out['reason_pred_rowsum_%d' % j] = r.detach().double().sum(1).numpy()
    if full or name == 'mid':
        out['log_prob'] = log_prob.detach().numpy()
    else:
        lp = log_prob.detach()
        top5 = lp.topk(5, dim=2)
        out['log_prob_top5_val'] = top5.values.numpy()
        out['log_prob_top5_idx'] = top5.indices.numpy()
        out['log_prob_target'] = lp.gather(2, labels[:, 1:1 + lp.size(1)].unsqueeze(2)).squeeze(2).numpy()
    for k in sorted(grads):
        n, s = grad_summary(k, grads[k])
        out['gradnorm/' + k] = n
        out['gradslice/' + k] = s
        n, s = grad_summary(k, stepped[k] - P[k])
        out['stepnorm/' + k] = n
        if full:
            out['grad/' + k] = grads[k].numpy()
            out['stepped/' + k] = stepped[k].numpy()

    # label smoothing variant (misc/utils.py:166-177)
    cfg_ls = cfg_of(spec)
    cfg_ls.use_label_smoothing = 1
    crit_ls = ref_utils.ReviewNetEnsembleCriterion(cfg_ls)
    model.zero_grad()
    log_prob2, top_pred2 = model(fc, att, labels)
    loss_ls = crit_ls(log_prob2, labels[:, 1:], masks[:, 1:], top_pred2, top, 1.0)
    o_ls = O.xe_criterion(cfg_ls, o_lp, labels[:, 1:], masks[:, 1:], o_rp, top, 1.0)
    close(o_ls, loss_ls, 1e-4, 'xe loss (label smoothing)')
    out['xe_loss_ls'] = np.float64(loss_ls.item())
    if full or name == 'mid':
        loss_ls.backward()
        for k, p in model.named_parameters():
            n, s = grad_summary(k, p.grad)
            out['gradnorm_ls/' + k] = n
    model.zero_grad()

    # ---- greedy sample (misc/RecurrentFusionModel.py:545-658) ---------------------------------
    with torch.no_grad():
        seq, seq_lp, lp_all, rp = model.sample(fc, att, {'sample_max': 1, 'beam_size': 1})
        o_seq, o_seq_lp, o_lp_all, o_rp2 = O.sample_greedy(cfg, P, fc, att)
    assert torch.equal(seq, o_seq), 'greedy ids differ'
    close(o_seq_lp, seq_lp, 2e-5, 'seqLogprobs')
    close(o_lp_all, lp_all, 2e-5, 'logprobs_all')
    out['greedy_seq'] = seq.numpy()
    out['greedy_seq_logprobs'] = seq_lp.numpy()
    top2 = lp_all.topk(2, dim=2).values
    out['greedy_margin'] = (top2[:, :, 0] - top2[:, :, 1]).numpy()
    out['greedy_logprobs_all_shape'] = np.array(lp_all.shape)
    if full or name == 'mid':
        out['greedy_logprobs_all'] = lp_all.numpy()
    else:
        t5 = lp_all.topk(5, dim=2)
        out['greedy_top5_val'] = t5.values.numpy()
        out['greedy_top5_idx'] = t5.indices.numpy()

    # ---- RL: multinomial sample with grad + reward criterion (train_rl.py:160-191) -------------
    if name in ('tiny0', 'tiny1', 'tinymax', 'odd', 'mid', 'c2'):
        rl_section(model, ref_utils, cfg, spec, P, fc, att, top, out)

    # ---- beam search (misc/RecurrentFusionModel.py:352-543) -----------------------------------
    if name in ('tiny0', 'tinymax', 'odd', 'mid', 'c2'):
        beam = 3
        nb = min(spec['B'], 3)
        fcb = [f[:nb] for f in fc]
        attb = [a[:nb] for a in att]
        with torch.no_grad(), LegacyIndexing():
            b_seq, b_lp, b_top_seq, b_top_prob, b_rp = model.sample(fcb, attb, {'beam_size': beam})
        with torch.no_grad():
            o_b = O.sample_beam(cfg, P, fcb, attb, beam)
        assert torch.equal(o_b[0], b_seq), 'beam seq differs'
        close(o_b[1], b_lp, 2e-5, 'beam seqLogprobs')
        for k in range(nb):
            assert torch.equal(o_b[2][k], b_top_seq[k]), 'beam top_seq differs'
        out['beam_size'] = np.int64(beam)
        out['beam_seq'] = b_seq.numpy()
        out['beam_seq_logprobs'] = b_lp.numpy()
        for k in range(nb):
            out['beam_top_seq_%d' % k] = b_top_seq[k].numpy()
            out['beam_top_prob_%d' % k] = np.array([float(x) for x in b_top_prob[k]], dtype=np.float64)

    # ---- single cells (a1, a2, a4, a5) on fresh random inputs ---------------------------------
    if full or name == 'mid':
        B, R, M = spec['B'], spec['R'], len(spec['feats'])
        rng = np.random.default_rng(4000 + spec['seed'])
        rnd = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32))  # noqa: E731
        with torch.no_grad():
            # a2: stage-I cell (t=1, encoder M-1)
            H, h, c = rnd(B, M * R), rnd(B, R), rnd(B, R)
            cell = model.review_steps_individual[1].lstm[M - 1]
            o, (nh, nc) = cell(H, att[M - 1], (h.unsqueeze(0), c.unsqueeze(0)))
            oh, oc, aux = O.fusion_cell(H, att[M - 1], h, c, P, 'review_steps_individual.1.lstm.%d.' % (M - 1), R)
            close(oh, o, 1e-5, 'a2 h')
            close(oc, nc[0], 1e-5, 'a2 c')
            z = cell.att_model(h, att[M - 1])
            close(aux['z'], z, 1e-5, 'a1 z')
            out['cell_a2_H'], out['cell_a2_h'], out['cell_a2_c'] = H.numpy(), h.numpy(), c.numpy()
            out['cell_a2_out_h'], out['cell_a2_out_c'] = o.numpy(), nc[0].numpy()
            out['cell_a1_z'], out['cell_a1_alpha'] = z.numpy(), aux['alpha'].numpy()
            # a4: stage-II cell (t=2)
            th = [rnd(B, spec['T1'], R) for _ in range(M)]
            o4, (nh4, nc4) = model.review_steps[2](th, (h.unsqueeze(0), c.unsqueeze(0)))
            oh4, oc4, _ = O.review_cell(th, h, c, P, 2, R, cfg.review_maxout)
            close(oh4, o4, 1e-5, 'a4 h')
            for i in range(M):
                out['cell_a4_thoughts_%d' % i] = th[i].numpy()
            out['cell_a4_out_h'], out['cell_a4_out_c'] = o4.numpy(), nc4[0].numpy()
            # a5: decoder cell
            xt, comb = rnd(B, cfg.input_encoding_size), rnd(B, spec['T2'], R)
            o5, (nh5, nc5) = model.decoder(xt, comb, (h.unsqueeze(0), c.unsqueeze(0)))
            oh5, oc5, _ = O.decoder_cell(xt, comb, h, c, P, R, cfg.maxout)
            close(oh5, o5, 1e-5, 'a5 h')
            out['cell_a5_xt'], out['cell_a5_comb'] = xt.numpy(), comb.numpy()
            out['cell_a5_out_h'], out['cell_a5_out_c'] = o5.numpy(), nc5[0].numpy()

    path = os.path.join(outdir, name + '.npz')
    np.savez_compressed(path, **out)
    print('%-6s log_prob err %.2e  xe %.6f  greedy T=%d  -> %s (%.1f KB)' % (
        name, e1, loss.item(), seq.size(1), path, os.path.getsize(path) / 1024))


def generate_decode(name, RefModel, ref_utils, outdir):
    """Decode tiers: the eval loop body (eval_utils.py:149-151, 159-208) with greedy and with beam search, and the
    self-critical sample path (train_rl.py:160-191, get_rewards.py:119-126) at the config's shape."""
    spec = CONFIGS[name]
    cfg = cfg_of(spec)
    spi, beam = spec['decode']['spi'], spec['decode']['beam']
    torch.manual_seed(0)
    model = RefModel(cfg)
    P = O.seeded_params(cfg, spec['seed'])
    model.load_state_dict(P)
    model.eval()
    fc0, att0, labels, masks, top = batch_of(cfg, spec)
    fc, att, _, _, _ = caption_rows(cfg, spec)
    out = dict(weights_digest=digest([P[k] for k in sorted(P)]), inputs_digest=digest(fc0 + att0),
               labels=labels.numpy(), masks=masks.numpy(), top_words=top.numpy(), seq_per_img=np.int64(spi),
               beam_size=np.int64(beam))
    crit = ref_utils.ReviewNetEnsembleCriterion(cfg)
    rows = np.arange(spec['B'] // spi) * spi
    with torch.no_grad():
        log_prob, top_pred = model(fc, att, labels)                                   # eval_utils.py:149-151
        loss = crit(log_prob, labels[:, 1:], masks[:, 1:], top_pred, top, 1.0)
        fc_u, att_u = [f[rows] for f in fc], [a[rows] for a in att]                   # :172-173
        g = model.sample(fc_u, att_u, {'beam_size': 1, 'sample_max': 1})             # :195
        g_sent = torch.sum(g[1] * (g[0] > 0).float(), 1)                             # :207
        with LegacyIndexing():
            b = model.sample(fc_u, att_u, {'beam_size': beam, 'sample_max': 1})
        b_sent = torch.sum(b[1] * (b[0] > 0).float(), 1)
    o_loss, o_seq, o_lp, o_sent = O.eval_step(cfg, P, fc, att, labels, masks, top, spi, 1.0, 1)
    close(o_loss, loss, 1e-4, 'eval xe loss')
    assert torch.equal(o_seq, g[0]), 'eval greedy ids differ'
    close(o_lp, g[1], 2e-5, 'eval greedy seqLogprobs')
    close(o_sent, g_sent, 1e-4, 'eval greedy sentence log-prob')
    ob_loss, ob_seq, ob_lp, ob_sent = O.eval_step(cfg, P, fc, att, labels, masks, top, spi, 1.0, beam)
    assert torch.equal(ob_seq, b[0]), 'eval beam ids differ'
    close(ob_lp, b[1], 2e-5, 'eval beam seqLogprobs')
    close(ob_sent, b_sent, 1e-4, 'eval beam sentence log-prob')
    with torch.no_grad():
        o_b = O.sample_beam(cfg, P, fc_u, att_u, beam)
    nb = len(rows)
    for k in range(nb):
        assert torch.equal(o_b[2][k], b[2][k]), 'beam top_seq differs'
    out['eval_xe_loss'] = np.float64(loss.item())
    out['eval_greedy_seq'], out['eval_greedy_seq_logprobs'] = g[0].numpy(), g[1].numpy()
    out['eval_greedy_sentence'] = g_sent.numpy()
    top2 = g[2].topk(2, dim=2).values
    out['eval_greedy_margin'] = (top2[:, :, 0] - top2[:, :, 1]).numpy()
    out['beam_seq'], out['beam_seq_logprobs'], out['beam_sentence'] = b[0].numpy(), b[1].numpy(), b_sent.numpy()
    for k in range(nb):
        out['beam_top_seq_%d' % k] = b[2][k].numpy()
        out['beam_top_prob_%d' % k] = np.array([float(x) for x in b[3][k]], dtype=np.float64)
    # the self-critical sample path on the image rows
    top_u = top[rows]
    rl_section(model, ref_utils, cfg, dict(spec, B=nb), P, fc_u, att_u, top_u, out)
    # get_rewards.py:119-126: the greedy baseline of the same batch is the eval greedy sample above
    path = os.path.join(outdir, name + '.npz')
    np.savez_compressed(path, **out)
    print('%-7s eval xe %.6f  greedy T=%d  beam=%d T=%d  rl loss %.6f -> %s (%.1f KB)' % (
        name, loss.item(), g[0].size(1), beam, b[0].size(1), float(out['rl_loss']), path, os.path.getsize(path) / 1024))


DROP_PROBS = dict(drop_prob_fusion=0.1, drop_prob_reason=0.2, drop_prob_lm=0.3)
DROP_TIERS = ('tiny0', 'tinymax', 'mid')


def generate_dropout(name, RefModel, ref_utils, outdir):
    """Training-mode parity: the reference in train() mode with drop_prob_fusion / _reason / _lm =
    0.1 / 0.2 / 0.3 and forward hooks on every nn.Dropout it owns, capturing each call's keep mask in call order
    (misc/RecurrentFusionModel.py:70, misc/LSTMSoftMultiAttentionFeatArrayNoInputCore.py:69,
    misc/LSTMSoftAttentionCore.py:98).  The oracle run with exactly those masks must reproduce log-probs, reason heads,
    loss and every gradient; masks and results are stored as `<name>_drop.npz`."""
    spec = dict(CONFIGS[name])
    spec['extra'] = dict(spec.get('extra', {}), **DROP_PROBS)
    cfg = cfg_of(spec)
    torch.manual_seed(0)
    model = RefModel(cfg)
    P = O.seeded_params(cfg, spec['seed'])
    model.load_state_dict(P)
    model.train()
    model.ss_prob = 0.0
    fc, att, labels, masks, top = batch_of(cfg, spec)
    M, T1, T2 = len(spec['feats']), spec['T1'], spec['T2']
    calls = dict(fusion={}, review={}, decoder=[])

    def capture(store, key):
        def hook(mod, inp, outp):
            keep = (outp != 0) | (inp[0] == 0)        # a kept unit is scaled, never zeroed (an exact 0 input: either way)
            scaled = inp[0] * (keep.to(inp[0].dtype) * (1.0 / (1.0 - mod.p)))
            assert float((scaled - outp).abs().max()) <= 1e-6 * max(1.0, float(outp.abs().max())), 'not inverted dropout'
            if key is None:
                store.append(keep.clone())
            else:
                assert key not in store, 'dropout site called twice'
                store[key] = keep.clone()
        return hook

    hooks = []
    for t in range(T1):
        for i in range(M):
            cell = model.review_steps_individual[t].lstm[i]
            assert abs(cell.dropout.p - cfg.drop_prob_fusion) < 1e-12
            hooks.append(cell.dropout.register_forward_hook(capture(calls['fusion'], (t, i))))
        # FeatArrayFusionNoInputCore owns a second nn.Dropout (:99) that its forward never calls
        hooks.append(model.review_steps_individual[t].dropout.register_forward_hook(
            lambda *a: (_ for _ in ()).throw(AssertionError('FeatArrayFusionNoInputCore.dropout was called'))))
    for t in range(T2):
        assert abs(model.review_steps[t].dropout.p - cfg.drop_prob_reason) < 1e-12
        hooks.append(model.review_steps[t].dropout.register_forward_hook(capture(calls['review'], t)))
    assert abs(model.decoder.dropout.p - cfg.drop_prob_lm) < 1e-12
    hooks.append(model.decoder.dropout.register_forward_hook(capture(calls['decoder'], None)))

    crit = ref_utils.ReviewNetEnsembleCriterion(cfg)
    torch.manual_seed(500 + spec['seed'])
    model.zero_grad()
    log_prob, top_pred = model(fc, att, labels)
    loss = crit(log_prob, labels[:, 1:], masks[:, 1:], top_pred, top, 1.0)
    loss.backward()
    for h in hooks:
        h.remove()
    S = log_prob.size(1)
    assert len(calls['fusion']) == T1 * M and len(calls['review']) == T2 and len(calls['decoder']) == S
    drop = O.make_drop(cfg, [[calls['fusion'][(t, i)] for i in range(M)] for t in range(T1)],
                       [calls['review'][t] for t in range(T2)], calls['decoder'])
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    o_lp, o_rp = O.forward(cfg, P, fc, att, labels, drop=drop)
    e1 = close(o_lp, log_prob, 2e-5, 'log_prob (dropout)')
    for a, b in zip(o_rp, top_pred):
        close(a, b, 2e-5, 'reason_pred (dropout)')
    o_loss, o_grads = O.train_step_loss_and_grads(cfg, P, fc, att, labels, masks, top, 1.0, drop=drop)
    close(o_loss, loss, 1e-4, 'xe loss (dropout)')
    for k in grads:
        close(o_grads[k], grads[k], 2e-5 + 1e-4 * float(grads[k].abs().max()), 'grad (dropout) ' + k)
    # the masks matter: the eval-mode oracle is far from these numbers
    assert float((O.forward(cfg, P, fc, att, labels)[0] - log_prob).abs().max()) > 1e-3
    full = name.startswith('tiny')
    out = dict(weights_digest=digest([P[k] for k in sorted(P)]), inputs_digest=digest(fc + att), labels=labels.numpy(),
               masks=masks.numpy(), top_words=top.numpy(),
               drop_probs=np.array([cfg.drop_prob_fusion, cfg.drop_prob_reason, cfg.drop_prob_lm]),
               keep_fusion=np.stack([np.stack([calls['fusion'][(t, i)].numpy() for i in range(M)]) for t in range(T1)]),
               keep_review=np.stack([calls['review'][t].numpy() for t in range(T2)]),
               keep_decoder=np.stack([m.numpy() for m in calls['decoder']]),
               log_prob=log_prob.detach().numpy(), xe_loss=np.float64(loss.item()))
    for j, r in enumerate(top_pred):
        out['reason_pred_%d' % j] = r.detach().numpy()
    for k in sorted(grads):
        n, sl = grad_summary(k, grads[k])
        out['gradnorm/' + k] = n
        out['gradslice/' + k] = sl
        if full:
            out['grad/' + k] = grads[k].numpy()
    path = os.path.join(outdir, name + '_drop.npz')
    np.savez_compressed(path, **out)
    kept = [float(np.mean(out[k])) for k in ('keep_fusion', 'keep_review', 'keep_decoder')]
    print('%-12s log_prob err %.2e  xe %.6f  keep rates %.3f %.3f %.3f -> %s (%.1f KB)' % (
        name + '_drop', e1, loss.item(), kept[0], kept[1], kept[2], path, os.path.getsize(path) / 1024))


SHOWTELL = dict(fc=2048, R=512, V=9487, B=4, S=16, seed=11, max_words=9)   # BASELINE config 1
SHOWTELL_END_BIAS = 0.5     # added to logit.bias[0] for the second beam-search fixture


def showtell_cfg():
    from types import SimpleNamespace
    c = SHOWTELL
    return SimpleNamespace(vocab_size=c['V'], input_encoding_size=c['R'], rnn_type='lstm', rnn_size=c['R'], num_layers=1,
                           drop_prob_lm=0.0, seq_length=c['S'], fc_feat_size=c['fc'], use_cuda=0, use_label_smoothing=0,
                           label_smoothing_epsilon=0.1, caption_model='show_tell')


def showtell_weights(model, seed):
    """our documented stream: uniform(+-0.1) over sorted(state_dict keys) from np.random.default_rng(seed)"""
    rng = np.random.default_rng(seed)
    sd = model.state_dict()
    return {k: torch.from_numpy(rng.uniform(-0.1, 0.1, tuple(sd[k].shape)).astype(np.float32)) for k in sorted(sd)}


def showtell_batch():
    c = SHOWTELL
    rng = np.random.default_rng(1000 + c['seed'])
    fc = torch.from_numpy(rng.standard_normal((c['B'], c['fc'])).astype(np.float32))
    labels = torch.zeros(c['B'], c['S'] + 2, dtype=torch.long)
    masks = torch.zeros(c['B'], c['S'] + 2)
    for b in range(c['B']):
        n = int(rng.integers(1, c['max_words'] + 1))
        labels[b, 1:1 + n] = torch.from_numpy(rng.integers(1, c['V'] + 1, n))
        masks[b, :n + 2] = 1
    return fc, labels, masks


def generate_showtell(outdir):
    """BASELINE config 1 (ShowTellModel single-encoder greedy decode on CPU, B=4, 2048-d feats, seq_len 16): the
    reference's forward + LanguageModelCriterion + greedy sample on seeded weights."""
    sys.path.insert(0, REF)
    from misc.ShowTellModel import ShowTellModel as RefShowTell  # noqa
    import misc.utils as ref_utils  # noqa
    cfg = showtell_cfg()
    torch.manual_seed(0)
    model = RefShowTell(cfg)
    W = showtell_weights(model, SHOWTELL['seed'])
    model.load_state_dict(W)
    model.eval()
    fc, labels, masks = showtell_batch()
    with torch.no_grad():
        lp = model(fc, None, labels)
        loss = ref_utils.LanguageModelCriterion(cfg)(lp, labels[:, 1:], masks[:, 1:])
        cfg.use_label_smoothing = 1
        loss_ls = ref_utils.LanguageModelCriterion(cfg)(lp, labels[:, 1:], masks[:, 1:])
        seq, seq_lp, lp_all = model.sample(fc, None, {'sample_max': 1})
    # beam search (misc/ShowTellModel.py:95-185) needs PyTorch-0.3.1 indexing like the fusion model's; twice: on the seeded
    # weights (no beam ends early) and with the END logit raised by SHOWTELL_END_BIAS, where beams that emitted END keep
    # competing and the done lists grow past the beam size
    beams = {}
    for tag, bias in (('beam3', 0.0), ('beam3e', SHOWTELL_END_BIAS)):
        Wb = dict(W)
        Wb['logit.bias'] = W['logit.bias'].clone()
        Wb['logit.bias'][0] += bias
        model.load_state_dict(Wb)
        with torch.no_grad(), LegacyIndexing():
            bseq, blp = model.sample(fc, None, {'beam_size': 3})
        counts = [len(d) for d in model.done_beams]
        pad = max(counts)
        beams[tag + '_seq'] = bseq.numpy().copy()
        beams[tag + '_seq_logprobs'] = blp.numpy().copy()
        beams[tag + '_done_counts'] = np.array(counts)
        beams[tag + '_done_p'] = np.array([[float(b['p']) for b in d] + [0.0] * (pad - len(d)) for d in model.done_beams])
        beams[tag + '_done_seq'] = np.stack([torch.stack([b['seq'] for b in d] + [torch.zeros_like(d[0]['seq'])] * (pad - len(d))).numpy()
                                             for d in model.done_beams])
    model.load_state_dict(W)
    t5 = lp.topk(5, dim=2)
    out = dict(weights_digest=digest([W[k] for k in sorted(W)]), inputs_digest=digest([fc]), labels=labels.numpy(),
               log_prob_shape=np.array(lp.shape), log_prob_top5_val=t5.values.numpy(), log_prob_top5_idx=t5.indices.numpy(),
               log_prob_target=lp.gather(2, labels[:, 1:1 + lp.size(1)].unsqueeze(2)).squeeze(2).numpy(),
               xe_loss=np.float64(loss.item()), xe_loss_ls=np.float64(loss_ls.item()),
               greedy_seq=seq.numpy(), greedy_seq_logprobs=seq_lp.numpy(), greedy_logprobs_all_shape=np.array(lp_all.shape),
               state_dict_keys=np.array(sorted(W)))
    out.update(beams)
    path = os.path.join(outdir, 'showtell.npz')
    np.savez_compressed(path, **out)
    print('showtell xe %.6f greedy T=%d -> %s (%.1f KB)' % (loss.item(), seq.size(1), path, os.path.getsize(path) / 1024))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default='tiny0,tiny1,tinymax,odd,mid,c2,c3,c5,evalmid,showtell,tiny0_drop,tinymax_drop,mid_drop')
    ap.add_argument('--out', default=os.path.join(ROOT, 'tests', 'golden'))
    args = ap.parse_args()
    torch.set_num_threads(8)
    RefModel, ref_utils = load_reference()
    os.makedirs(args.out, exist_ok=True)
    for name in args.only.split(','):
        if name == 'showtell':
            generate_showtell(args.out)
            continue
        if name.endswith('_drop'):
            generate_dropout(name[:-5], RefModel, ref_utils, args.out)
            continue
        if 'decode' in CONFIGS[name]:
            generate_decode(name, RefModel, ref_utils, args.out)
        else:
            generate(name, RefModel, ref_utils, args.out)


if __name__ == '__main__':
    main()
