"""CPU oracle for the recurrent-fusion caption decoder hot path.

TEST INFRASTRUCTURE ONLY.  This file is a plain PyTorch-CPU restatement of the
reference algorithm; it exists to check the HIP path and to time a CPU baseline.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it.  The product (``recurrent_fusion_network_amd``) never does.

Parity pin: the reference ships no tests or golden vectors for this path
(SURVEY.md section 4), so this oracle is pinned against outputs of the reference
itself, run in the build container by ``oracle/make_golden.py`` and committed
under ``tests/golden/`` (see ``tests/test_oracle_golden.py``).

Every function cites the reference lines it restates (paths relative to the
reference checkout).  Parameters are passed as a flat ``dict`` whose keys are
the reference's ``state_dict`` names.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch

Tensor = torch.Tensor


# --------------------------------------------------------------------------- #
# configuration / parameter schema
# --------------------------------------------------------------------------- #
def make_cfg(feat_array_info: Sequence[dict], vocab_size: int, rnn_size: int = 512,
             input_encoding_size: int = 512, att_hid_size: int = 512,
             num_review_steps_0: int = 8, num_review_steps: int = 8,
             top_words_count: int = 1000, seq_length: int = 16, **extra) -> SimpleNamespace:
    """The Namespace fields RecurrentFusionModel.__init__ reads (misc/RecurrentFusionModel.py:118-151)."""
    cfg = SimpleNamespace(
        vocab_size=vocab_size, input_encoding_size=input_encoding_size, rnn_type='lstm',
        rnn_size=rnn_size, num_layers=1, drop_prob_lm=0.0, drop_prob_reason=0.0,
        drop_prob_fusion=0.0, seq_length=seq_length, num_review_steps=num_review_steps,
        num_review_steps_0=num_review_steps_0, top_words_count=top_words_count,
        att_hid_size=att_hid_size, review_maxout=0, maxout=0, fusion_maxout=0, use_cuda=0,
        feat_array_info=[dict(d) for d in feat_array_info],
        caption_model='recurrent_fusion_model',
        use_label_smoothing=0, label_smoothing_epsilon=0.1, use_ppo=0, ppo_clip=0.2,
    )
    for k, v in extra.items():
        setattr(cfg, k, v)
    return cfg


def param_shapes(cfg) -> Dict[str, Tuple[int, ...]]:
    """state_dict schema of RecurrentFusionModel (misc/RecurrentFusionModel.py:153-184,
    misc/AttentionModelCore.py:16-18, misc/LSTMSoftMultiAttentionFeatArrayNoInputCore.py:24-32,
    misc/LSTMSoftAttentionCore.py:24-37).  nn.Linear layout is (out, in)."""
    M = len(cfg.feat_array_info)
    R, A, E = cfg.rnn_size, cfg.att_hid_size, cfg.input_encoding_size
    V1, K = cfg.vocab_size + 1, cfg.top_words_count
    T1, T2 = cfg.num_review_steps_0, cfg.num_review_steps
    g2 = (5 if cfg.review_maxout else 4) * R
    gd = (5 if cfg.maxout else 4) * R
    s: Dict[str, Tuple[int, ...]] = {}

    def lin(name, out, inp):
        s[name + '.weight'] = (out, inp)
        s[name + '.bias'] = (out,)

    def att(prefix, feat):
        lin(prefix + 'att_2_att_h', A, feat)
        lin(prefix + 'h_2_att_h', A, R)
        lin(prefix + 'att_h_2_out', 1, A)

    for i, info in enumerate(cfg.feat_array_info):
        lin('fc2h.%d' % i, R, info['fc_feat_size'])
    s['embed.weight'] = (V1, E)
    lin('logit', V1, R)
    for t in range(T1):
        for i, info in enumerate(cfg.feat_array_info):
            p = 'review_steps_individual.%d.lstm.%d.' % (t, i)
            att(p + 'att_model.', info['att_feat_size'])
            lin(p + 'H2h', 4 * R, M * R)            # fusion_maxout is never forwarded (:94-96)
            lin(p + 'z2h', 4 * R, info['att_feat_size'])
    for i in range(M):
        lin('reason_linear_individual.%d' % i, K, R)
    for t in range(T2):
        p = 'review_steps.%d.' % t
        lin(p + 'h2h', g2, R)
        for i in range(M):
            lin(p + 'z_2_h.%d' % i, g2, R)
        for i in range(M):
            att(p + 'att_model.%d.' % i, R)
    lin('reason_linear', K, R)
    lin('decoder.i2h', gd, E)
    lin('decoder.h2h', gd, R)
    lin('decoder.z2h', gd, R)
    att('decoder.', R)
    return s


def seeded_params(cfg, seed: int, scale: float = 0.1, dtype=torch.float32) -> Dict[str, Tensor]:
    """Documented weight stream shared by the golden generator, the tests and bench.py:
    np.random.default_rng(seed), uniform(-scale, scale) float32, iterating sorted(keys)."""
    rng = np.random.default_rng(seed)
    shapes = param_shapes(cfg)
    out = {}
    for k in sorted(shapes):
        a = rng.uniform(-scale, scale, size=shapes[k]).astype(np.float32)
        out[k] = torch.from_numpy(a).to(dtype)
    return out


def synthetic_batch(cfg, B: int, seed: int, n_words: int = None, dtype=torch.float32):
    """Synthetic inputs of SURVEY.md section 8d: N(0,1) features, full-length captions
    (col 0 = BOS 0, cols 1..seq_length uniform in [1, V], last col 0), masks all one,
    5 distinct top-word targets per row then -1 padding (dataloader.py:312-332)."""
    rng = np.random.default_rng(seed)
    fc, att = [], []
    for info in cfg.feat_array_info:
        fc.append(torch.from_numpy(rng.standard_normal((B, info['fc_feat_size'])).astype(np.float32)).to(dtype))
    for info in cfg.feat_array_info:
        att.append(torch.from_numpy(
            rng.standard_normal((B, info['att_num'], info['att_feat_size'])).astype(np.float32)).to(dtype))
    S = cfg.seq_length
    n_words = S if n_words is None else n_words
    labels = np.zeros((B, S + 2), dtype=np.int64)
    labels[:, 1:1 + n_words] = rng.integers(1, cfg.vocab_size + 1, size=(B, n_words))
    masks = np.zeros((B, S + 2), dtype=np.float32)
    masks[:, :n_words + 2] = 1.0
    K = cfg.top_words_count
    top = -np.ones((B, K), dtype=np.int64)
    n_top = min(5, K)
    for b in range(B):
        top[b, :n_top] = rng.choice(K, size=n_top, replace=False)
    return fc, att, torch.from_numpy(labels), torch.from_numpy(masks), torch.from_numpy(top)


# --------------------------------------------------------------------------- #
# cells
# --------------------------------------------------------------------------- #
def attention(pre_h: Tensor, att_seq: Tensor, P: Dict[str, Tensor], prefix: str):
    """Additive soft attention (misc/AttentionModelCore.py:31-48; inlined copy in
    misc/LSTMSoftAttentionCore.py:64-79).  Returns (z, alpha, scores)."""
    Wa, ba = P[prefix + 'att_2_att_h.weight'], P[prefix + 'att_2_att_h.bias']
    Wh, bh = P[prefix + 'h_2_att_h.weight'], P[prefix + 'h_2_att_h.bias']
    wo, bo = P[prefix + 'att_h_2_out.weight'], P[prefix + 'att_h_2_out.bias']
    att_linear = att_seq @ Wa.t() + ba                       # :32-34  (B, L, A)
    h_linear = pre_h @ Wh.t() + bh                           # :36     (B, A)
    att_h = torch.tanh(h_linear.unsqueeze(1) + att_linear)   # :37-39
    scores = att_h @ wo.view(-1) + bo                        # :41-43  (B, L)
    alpha = torch.softmax(scores, dim=1)                     # :44
    z = torch.bmm(att_seq.transpose(1, 2), alpha.unsqueeze(2)).squeeze(2)   # :45-47 (B, D)
    return z, alpha, scores


def apply_dropout(next_h: Tensor, keep: Tensor = None, p: float = 0.0) -> Tensor:
    """nn.Dropout in training mode with the Bernoulli draw given explicitly (misc/RecurrentFusionModel.py:70,
    misc/LSTMSoftMultiAttentionFeatArrayNoInputCore.py:69, misc/LSTMSoftAttentionCore.py:98): kept units are scaled by
    1 / (1 - p), dropped ones are 0.  It acts on next_h only: the recurrent h, every consumer of the cell output (reason
    heads, thought vectors, logit layer, the concatenated H) see the POST-dropout value; next_c is never dropped.
    ``keep`` None = eval mode."""
    if keep is None:
        return next_h
    return next_h * (keep.to(next_h.dtype) * (1.0 / (1.0 - p)))


def lstm_update(sums: Tensor, pre_c: Tensor, R: int, maxout: int = 0):
    """Gate epilogue shared by the three cells (misc/RecurrentFusionModel.py:55-68):
    chunk order [in | forget | out | g], NOT cuDNN's."""
    sig = torch.sigmoid(sums[:, :3 * R])
    in_gate, forget_gate, out_gate = sig[:, :R], sig[:, R:2 * R], sig[:, 2 * R:3 * R]
    if maxout:
        g = torch.max(sums[:, 3 * R:4 * R], sums[:, 4 * R:5 * R])
    else:
        g = torch.tanh(sums[:, 3 * R:4 * R])
    next_c = forget_gate * pre_c + in_gate * g
    next_h = out_gate * torch.tanh(next_c)
    return next_h, next_c


def fusion_cell(H: Tensor, att_feat: Tensor, h: Tensor, c: Tensor, P, prefix: str, R: int, keep: Tensor = None,
                p: float = 0.0):
    """Stage-I cell (misc/RecurrentFusionModel.py:47-74); ``keep``: this call's dropout mask (None = eval)."""
    z, alpha, _ = attention(h, att_feat, P, prefix + 'att_model.')
    sums = (H @ P[prefix + 'H2h.weight'].t() + P[prefix + 'H2h.bias']
            + z @ P[prefix + 'z2h.weight'].t() + P[prefix + 'z2h.bias'])          # :53
    nh, nc = lstm_update(sums, c, R, 0)
    nh = apply_dropout(nh, keep, p)                                               # :70
    return nh, nc, dict(z=z, alpha=alpha, sums=sums)


def fusion_step(att_feats: List[Tensor], hs: List[Tensor], cs: List[Tensor], P, t: int, R: int, drop=None):
    """One stage-I step over all encoders (misc/RecurrentFusionModel.py:101-114): H is built from
    the PREVIOUS step's (post-dropout) hidden states before any cell runs."""
    H = torch.cat(hs, 1)                                                          # :102-107
    nh, nc = [], []
    for i in range(len(hs)):
        a, b, _ = fusion_cell(H, att_feats[i], hs[i], cs[i], P,
                              'review_steps_individual.%d.lstm.%d.' % (t, i), R,
                              None if drop is None else drop['fusion'][t][i], 0.0 if drop is None else drop['p_fusion'])
        nh.append(a)
        nc.append(b)
    return nh, nc


def review_cell(thoughts: List[Tensor], h: Tensor, c: Tensor, P, t: int, R: int, maxout: int = 0, keep: Tensor = None,
                drop_p: float = 0.0):
    """Stage-II cell (misc/LSTMSoftMultiAttentionFeatArrayNoInputCore.py:41-73); ``keep``: dropout mask (None = eval)."""
    p = 'review_steps.%d.' % t
    sums = h @ P[p + 'h2h.weight'].t() + P[p + 'h2h.bias']                        # :50
    aux = []
    for i in range(len(thoughts)):
        z, alpha, _ = attention(h, thoughts[i], P, p + 'att_model.%d.' % i)        # :46-48
        sums = sums + z @ P[p + 'z_2_h.%d.weight' % i].t() + P[p + 'z_2_h.%d.bias' % i]   # :51-52
        aux.append(dict(z=z, alpha=alpha))
    nh, nc = lstm_update(sums, c, R, maxout)
    nh = apply_dropout(nh, keep, drop_p)                                          # :69
    return nh, nc, dict(sums=sums, att=aux)


def decoder_cell(xt: Tensor, comb: Tensor, h: Tensor, c: Tensor, P, R: int, maxout: int = 0, keep: Tensor = None,
                 p: float = 0.0):
    """Decoder cell (misc/LSTMSoftAttentionCore.py:60-102); ``keep``: dropout mask (None = eval)."""
    z, alpha, _ = attention(h, comb, P, 'decoder.')                               # :64-79
    sums = (xt @ P['decoder.i2h.weight'].t() + P['decoder.i2h.bias']
            + h @ P['decoder.h2h.weight'].t() + P['decoder.h2h.bias']
            + z @ P['decoder.z2h.weight'].t() + P['decoder.z2h.bias'])            # :81
    nh, nc = lstm_update(sums, c, R, maxout)
    nh = apply_dropout(nh, keep, p)                                               # :98
    return nh, nc, dict(z=z, alpha=alpha, sums=sums)


def make_drop(cfg, fusion, review, decoder):
    """Training-mode dropout with explicit masks: ``fusion[t][i]``, ``review[t]``, ``decoder[s]`` are the (B, R) keep
    masks (bool / 0-1) of the stage-I cell (t, i), the stage-II cell t and the decoder step s, in the order the
    reference's nn.Dropout layers are called; probabilities: stage I ``drop_prob_fusion``
    (misc/RecurrentFusionModel.py:160), stage II ``drop_prob_reason`` (:172), decoder ``drop_prob_lm`` (:183)."""
    return dict(fusion=fusion, review=review, decoder=decoder, p_fusion=float(cfg.drop_prob_fusion),
                p_reason=float(cfg.drop_prob_reason), p_lm=float(cfg.drop_prob_lm))


# --------------------------------------------------------------------------- #
# model phases
# --------------------------------------------------------------------------- #
def init_state(cfg, P, fc_feats):
    """misc/RecurrentFusionModel.py:333-343 (get_init_state): h0 = fc2h(fc), c0 = h0.clone()."""
    hs, cs = [], []
    for i in range(len(cfg.feat_array_info)):
        h0 = fc_feats[i] @ P['fc2h.%d.weight' % i].t() + P['fc2h.%d.bias' % i]
        hs.append(h0)
        cs.append(h0.clone())
    return hs, cs


def thought_vectors(cfg, P, att_feats, hs, cs, want_aux: bool = False, drop=None):
    """Stages I+II (misc/RecurrentFusionModel.py:283-331, same code at :210-255).
    Returns (thought_vectors_comb (B,T2,R), reason_pred list[M+1] of (B,K), (h, c))."""
    M, R = len(cfg.feat_array_info), cfg.rnn_size
    T1, T2 = cfg.num_review_steps_0, cfg.num_review_steps
    outs = [[] for _ in range(M)]
    reason = [[] for _ in range(M)]
    for t in range(T1):                                                           # :287-291
        hs, cs = fusion_step(att_feats, hs, cs, P, t, R, drop)
        for j in range(M):
            outs[j].append(hs[j])
            reason[j].append(hs[j] @ P['reason_linear_individual.%d.weight' % j].t()
                             + P['reason_linear_individual.%d.bias' % j])
    thoughts, reason_pred = [], []
    for i in range(M):                                                            # :295-305
        thoughts.append(torch.stack(outs[i]).transpose(0, 1).contiguous())        # (B,T1,R)
        reason_pred.append(torch.stack(reason[i]).transpose(0, 1).max(1)[0])
    h = sum(hs) / M                                                               # :307-309
    c = sum(cs) / M
    comb, reason_c = [], []
    for t in range(T2):                                                           # :315-318
        h, c, _ = review_cell(thoughts, h, c, P, t, R, cfg.review_maxout,
                              None if drop is None else drop['review'][t], 0.0 if drop is None else drop['p_reason'])
        comb.append(h)
        reason_c.append(h @ P['reason_linear.weight'].t() + P['reason_linear.bias'])
    comb_t = torch.stack(comb).transpose(0, 1).contiguous()                       # (B,T2,R)
    reason_pred.append(torch.stack(reason_c).transpose(0, 1).max(1)[0])
    if want_aux:
        return comb_t, reason_pred, (h, c), dict(thoughts=thoughts)
    return comb_t, reason_pred, (h, c)


def one_time_step(cfg, P, xt, comb, h, c, keep=None, p: float = 0.0):
    """misc/RecurrentFusionModel.py:345-350: returns PRE-softmax logits (of the post-dropout h)."""
    h, c, _ = decoder_cell(xt, comb, h, c, P, cfg.rnn_size, cfg.maxout, keep, p)
    return h @ P['logit.weight'].t() + P['logit.bias'], h, c


def forward(cfg, P, fc_feats, att_feats, seq: Tensor, drop=None):
    """Teacher-forced XE pass (misc/RecurrentFusionModel.py:198-281) with ss_prob = 0.  ``drop`` (make_drop): training
    mode with the given dropout masks; None: eval mode."""
    hs, cs = init_state(cfg, P, fc_feats)
    comb, reason_pred, (h, c) = thought_vectors(cfg, P, att_feats, hs, cs, drop=drop)
    outputs = []
    for i in range(seq.size(1)):                                                  # :259
        if i >= 1 and int(seq[:, i].sum()) == 0:                                  # :274
            break
        xt = P['embed.weight'][seq[:, i]]                                         # :276
        logits, h, c = one_time_step(cfg, P, xt, comb, h, c, None if drop is None else drop['decoder'][i],
                                     0.0 if drop is None else drop['p_lm'])
        outputs.append(torch.log_softmax(logits, dim=1))                          # :278
    return torch.stack(outputs, 1).contiguous(), reason_pred                      # :281


def sample_greedy(cfg, P, fc_feats, att_feats, force_ids: Tensor = None):
    """Free-running decode (misc/RecurrentFusionModel.py:545-658), sample_max=1.
    ``force_ids`` (B, seq_length) replays externally drawn ids (multinomial path, :623-635).
    Returns (seq (B,<=16) int64, seqLogprobs, logprobs_all (B,<=17,V+1), reason_pred)."""
    B = fc_feats[0].size(0)
    hs, cs = init_state(cfg, P, fc_feats)
    comb, reason_pred, (h, c) = thought_vectors(cfg, P, att_feats, hs, cs)
    seq, seq_lp, lp_all = [], [], []
    logprobs = None
    unfinished = None
    for t in range(cfg.seq_length + 1):                                           # :616
        if t == 0:
            it = torch.zeros(B, dtype=torch.long)                                 # :618
        elif force_ids is None:
            sample_lp, it = torch.max(logprobs, 1)                                # :620
        else:
            it = force_ids[:, t - 1]
            sample_lp = logprobs.gather(1, it.view(-1, 1)).view(-1)               # :632
        xt = P['embed.weight'][it]                                                # :637 (unmasked ids)
        if t >= 1:
            unfinished = (it > 0) if t == 1 else unfinished & (it > 0)            # :641-644
            if int(unfinished.sum()) == 0:                                        # :645
                break
            seq.append(it * unfinished.long())                                    # :647-648
            seq_lp.append(sample_lp)                                              # :649 (NOT masked)
        logits, h, c = one_time_step(cfg, P, xt, comb, h, c)
        logprobs = torch.log_softmax(logits, dim=1)
        lp_all.append(logprobs)
    if len(seq) == 0:  # the reference would fail in torch.cat([]) here (:655); keep shapes sane
        return (torch.zeros(B, 0, dtype=torch.long), torch.zeros(B, 0), torch.stack(lp_all, 1), reason_pred)
    return torch.stack(seq, 1), torch.stack(seq_lp, 1), torch.stack(lp_all, 1).contiguous(), reason_pred


def sample_beam(cfg, P, fc_feats, att_feats, beam_size: int):
    """Per-image beam search (misc/RecurrentFusionModel.py:352-543).  Pure-Python bookkeeping,
    so only for small cases.  Candidate order: outer loop over sorted column c, inner over beam q
    (:470-478); stable sort by -p (:482); beams whose previous token was 0 are skipped (:475);
    only row 0 is active at t == 1 (:468-469).
    `P` may be a LIST of parameter dicts: the ensemble of eval_utils.eval_ensemble (eval_utils.py:387-720) -- every member
    keeps its own state, the members' logits of a step are summed and divided by their number before the log-softmax
    (eval_utils.py:268-317), everything else is the same search.  A single dict is a one-member list (x / 1 and a sum of
    one term leave every bit alone)."""
    members = list(P) if isinstance(P, (list, tuple)) else [P]
    B = fc_feats[0].size(0)
    S = cfg.seq_length
    seq = torch.zeros(S, B, dtype=torch.long)
    seq_lp = torch.zeros(S, B)
    top_seq, top_prob, reason_batch, done_all = [], [[] for _ in range(B)], [], []
    for k in range(B):
        fc_k = [f[k:k + 1].expand(beam_size, f.size(1)).contiguous() for f in fc_feats]   # :379-386
        att_k = [a[k:k + 1].expand(beam_size, a.size(1), a.size(2)).contiguous() for a in att_feats]
        st = []
        for Pm in members:
            hs, cs = init_state(cfg, Pm, fc_k)
            comb, reason_pred, (h, c) = thought_vectors(cfg, Pm, att_k, hs, cs)
            st.append([comb, h, c])
        reason_batch.append(reason_pred)
        beam_seq = torch.zeros(S, beam_size, dtype=torch.long)
        beam_lp = torch.zeros(S, beam_size)
        beam_sum = torch.zeros(beam_size)
        done = []
        logprobs = None
        for t in range(S + 1):                                                    # :451
            if t == 0:
                tok = torch.zeros(beam_size, dtype=torch.long)
            else:
                ys, ix = torch.sort(logprobs.float(), 1, True)                    # :463
                cands = []
                cols = min(beam_size, ys.size(1))
                rows = 1 if t == 1 else beam_size
                for cc in range(cols):
                    for q in range(rows):
                        local = float(ys[q, cc])
                        if t > 1 and int(beam_seq[t - 2, q]) == 0:                 # :475
                            continue
                        cands.append(dict(c=int(ix[q, cc]), q=q,
                                          p=float(np.float32(beam_sum[q]) + np.float32(local)), r=local))
                if len(cands) == 0:                                               # :480
                    break
                cands = sorted(cands, key=lambda x: -x['p'])                      # :482 (stable)
                new_st = [[m_[1].clone(), m_[2].clone()] for m_ in st]
                if t > 1:
                    prev_seq = beam_seq[:t - 1].clone()
                    prev_lp = beam_lp[:t - 1].clone()
                for vix in range(min(beam_size, len(cands))):                     # :491
                    v = cands[vix]
                    if t > 1:
                        beam_seq[:t - 1, vix] = prev_seq[:, v['q']]
                        beam_lp[:t - 1, vix] = prev_lp[:, v['q']]
                    for m_, n_ in zip(st, new_st):                                # :499-501
                        n_[0][vix] = m_[1][v['q']]
                        n_[1][vix] = m_[2][v['q']]
                    beam_seq[t - 1, vix] = v['c']
                    beam_lp[t - 1, vix] = v['r']
                    beam_sum[vix] = v['p']
                    if v['c'] == 0 or t == S:                                     # :508
                        done.append(dict(seq=beam_seq[:, vix].clone(), logps=beam_lp[:, vix].clone(),
                                         p=float(beam_sum[vix])))
                for m_, n_ in zip(st, new_st):
                    m_[1], m_[2] = n_
                tok = beam_seq[t - 1]
            total = None
            for Pm, m_ in zip(members, st):
                logits, m_[1], m_[2] = one_time_step(cfg, Pm, Pm['embed.weight'][tok], m_[0], m_[1], m_[2])
                total = logits if total is None else total + logits
            logprobs = torch.log_softmax(total / float(len(members)), dim=1)
        done = sorted(done, key=lambda x: -x['p'])                                # :529
        seq[:, k] = done[0]['seq']
        seq_lp[:, k] = done[0]['logps']
        cur = torch.zeros(len(done), S, dtype=torch.long)
        for j, d in enumerate(done):
            cur[j] = d['seq']
            top_prob[k].append(d['p'])
        top_seq.append(cur)
        done_all.append(done)
    return seq.t(), seq_lp.t(), top_seq, top_prob, reason_batch, done_all


# --------------------------------------------------------------------------- #
# criteria / optimiser
# --------------------------------------------------------------------------- #
def multilabel_margin(pred: Tensor, target: Tensor) -> Tensor:
    """nn.MultiLabelMarginLoss (mean reduction) written out: per row, targets are the ids before
    the first -1; loss = sum_{j in targets} sum_{i not in targets} max(0, 1 - (x[j] - x[i])) / K."""
    B, K = pred.shape
    total = pred.new_zeros(())
    for b in range(B):
        tg = []
        for j in range(K):
            if int(target[b, j]) < 0:
                break
            tg.append(int(target[b, j]))
        is_t = torch.zeros(K, dtype=torch.bool)
        if tg:
            is_t[torch.tensor(tg)] = True
        for j in tg:
            total = total + torch.clamp(1 - (pred[b, j] - pred[b][~is_t]), min=0).sum() / K
    return total / B


def xe_criterion(cfg, log_prob, target, mask, top_pred, top_true, reason_weight: float):
    """ReviewNetEnsembleCriterion.forward (misc/utils.py:161-192)."""
    B, T, V1 = log_prob.shape
    target = target[:, :T]
    mask = mask[:, :T]
    if cfg.use_label_smoothing:                                                   # :166-177
        eps = cfg.label_smoothing_epsilon
        one_hot = torch.zeros(B, T, V1, dtype=log_prob.dtype).scatter_(2, target.unsqueeze(2), 1.0)
        one_hot = one_hot * (1.0 - eps) + eps / V1
        out = (-(log_prob * one_hot).sum(2) * mask).sum() / B
    else:                                                                         # :179-184
        out = (-log_prob.gather(2, target.unsqueeze(2)).squeeze(2) * mask).sum() / B
    disc = sum(torch.nn.functional.multilabel_margin_loss(p, top_true) for p in top_pred)   # :186-190
    return out + disc * reason_weight / len(top_pred)


def rl_criterion(cfg, sample_logprobs, seq, reward, logprobs_all, entropy_reg, top_pred, top_true,
                 reason_weight, sample_logprobs_old=None):
    """ReviewNetRewardCriterion.forward (misc/utils.py:50-84)."""
    B, T = sample_logprobs.shape
    inp = sample_logprobs.contiguous().view(-1)
    reward = reward.contiguous().view(-1)
    mask_0 = (seq > 0).to(inp.dtype)
    mask = torch.cat([mask_0.new_ones(B, 1), mask_0[:, :-1]], 1).view(-1)          # :56-57
    lp = logprobs_all[:, :T, :]
    entropy_minus = (lp * torch.exp(lp)).sum(2) * mask_0                           # :59-61
    if cfg.use_ppo:                                                               # :62-69
        ratio = torch.exp(inp) / (1e-5 + torch.exp(sample_logprobs_old.contiguous().view(-1)))
        surr1 = ratio * reward
        surr2 = surr1.clamp(1 - cfg.ppo_clip, 1 + cfg.ppo_clip) * reward
        out = -torch.min(surr1, surr2) * mask
    else:
        out = -inp * reward * mask                                                # :71
    out = out.sum() / B + entropy_reg * entropy_minus.sum() / B                    # :72
    disc = sum(torch.nn.functional.multilabel_margin_loss(p, top_true) for p in top_pred)
    return out + disc * reason_weight / len(top_pred)                             # :78-82


def eval_step(cfg, P, fc_feats, att_feats, labels, masks, top_words, seq_per_img: int, reason_weight: float = 1.0,
              beam_size: int = 1):
    """One iteration of eval_split's loop (eval_utils.py:149-151, 159-208): XE loss on the caption batch (every
    image's features seq_per_img times in a row), then sample() on ONE row per image (rows arange(n) * seq_per_img)
    and the sentence score sum(seqLogprobs * (seq > 0)).  -> (loss, seq, seqLogprobs, log_probs_sentence)."""
    with torch.no_grad():
        log_prob, top_pred = forward(cfg, P, fc_feats, att_feats, labels)
        loss = xe_criterion(cfg, log_prob, labels[:, 1:], masks[:, 1:], top_pred, top_words, reason_weight)
        rows = torch.arange(fc_feats[0].size(0) // seq_per_img) * seq_per_img        # :172-173
        fc_u = [f[rows] for f in fc_feats]
        att_u = [a[rows] for a in att_feats]
        if beam_size > 1:
            out = sample_beam(cfg, P, fc_u, att_u, beam_size)
        else:
            out = sample_greedy(cfg, P, fc_u, att_u)
        seq, seq_lp = out[0], out[1]
        sentence = torch.sum(seq_lp * (seq > 0).to(seq_lp.dtype), 1)                   # :207
    return loss, seq, seq_lp, sentence


def clip_and_adam(params: Dict[str, Tensor], grads: Dict[str, Tensor], state: dict, lr=5e-4,
                  betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5, grad_clip=1.0):
    """clip_gradient (misc/utils.py:292-296: element-wise clamp) then torch.optim.Adam with L2
    weight decay folded into the gradient (train.py:69-71, 162-163).  In-place on ``params``."""
    state['step'] = state.get('step', 0) + 1
    t = state['step']
    for k, p in params.items():
        g = grads[k].clamp(-grad_clip, grad_clip)
        g = g + weight_decay * p
        m = state.setdefault('m', {}).setdefault(k, torch.zeros_like(p))
        v = state.setdefault('v', {}).setdefault(k, torch.zeros_like(p))
        m.mul_(betas[0]).add_(g, alpha=1 - betas[0])
        v.mul_(betas[1]).addcmul_(g, g, value=1 - betas[1])
        bc1 = 1 - betas[0] ** t
        bc2 = 1 - betas[1] ** t
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(m, denom, value=-lr / bc1)
    return params


def train_step_loss_and_grads(cfg, P: Dict[str, Tensor], fc, att, labels, masks, top_words,
                              reason_weight: float = 1.0, drop=None):
    """The reference's timed region minus the optimiser (train.py:143-160): forward, criterion,
    backward.  Returns (loss, grads dict)."""
    Pg = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
    log_prob, top_pred = forward(cfg, Pg, fc, att, labels, drop=drop)
    loss = xe_criterion(cfg, log_prob, labels[:, 1:], masks[:, 1:], top_pred, top_words, reason_weight)
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in Pg.items()}
    return loss.detach(), grads
