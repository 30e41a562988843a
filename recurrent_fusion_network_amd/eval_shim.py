"""The per-batch body of the reference's evaluation loop, for callers that port `eval_utils.eval_split`.

eval_utils.py:149-151 computes the XE loss on the caption batch (every image's features `seq_per_img` times in a
row, dataloader.py:251-252); :159-195 then keeps ONE row per image (`arange(batch) * seq_per_img`) and calls
`model.sample(..., {'beam_size', 'sample_max'})`; :206-208 scores each generated sentence as
`sum(seqLogprobs * (seq > 0))`.  Everything around it (vocabulary decoding, COCO json, Java metrics) is host-side
bookkeeping outside the accelerated path.

Plumbing only: the arithmetic is `model.forward` / `model.sample` / the criterion (HIP kernels).
"""
import torch


def unique_image_rows(n_rows, seq_per_img, device=None):
    """Row indices `np.arange(loader.batch_size) * loader.seq_per_img` (eval_utils.py:172-173)."""
    if n_rows % seq_per_img:
        raise ValueError('batch of %d rows is not a multiple of seq_per_img = %d' % (n_rows, seq_per_img))
    return torch.arange(n_rows // seq_per_img, device=device) * seq_per_img


def eval_step(model, crit, fc_feats, att_feats, labels, masks, top_words, seq_per_img, reason_weight=1.0,
              beam_size=1, sample_max=1):
    """-> dict(loss, seq, seqLogprobs, log_probs_sentence, sample): one iteration of eval_split's loop.

    fc_feats / att_feats: lists of caption-row tensors (each image repeated seq_per_img times), labels (rows, S+2),
    masks, top_words as the loader builds them.  `sample` is the full tuple model.sample returned (4 entries for
    beam_size 1, 5 with beam search: eval_utils.py:198-200)."""
    with torch.no_grad():
        log_prob, top_pred = model(fc_feats, att_feats, labels)
        loss = crit(log_prob, labels[:, 1:], masks[:, 1:], top_pred, top_words, reason_weight)
        rows = unique_image_rows(fc_feats[0].size(0), seq_per_img, fc_feats[0].device)
        fc_u = [f.index_select(0, rows) for f in fc_feats]
        att_u = [a.index_select(0, rows) for a in att_feats]
        out = model.sample(fc_u, att_u, {'beam_size': beam_size, 'sample_max': sample_max})
        seq, seq_lp = out[0], out[1]
        sentence = torch.sum(seq_lp * (seq > 0).to(seq_lp.dtype), 1)
    return dict(loss=loss, seq=seq, seqLogprobs=seq_lp, log_probs_sentence=sentence, sample=out)
