"""Ensemble decoding over the path's inference hooks.

Reference: eval_utils.model_ensemble_feat_array_one_step(_multi_gpu) (eval_utils.py:268-317) averages the members'
pre-softmax logits of one decoder step (sum, then / n), applies log_softmax, and every member continues with the
token chosen from the averaged distribution; the reference code around it is stale (it calls one_time_step /
get_thought_vectors with arities the model does not have) and moves logits between GPUs with
`.cuda()` copies.  Here the members' steps run through rfn_decoder_step, the logit sum / division / log-softmax /
greedy pick are HIP kernels, members living on other ranks contribute through ONE sum all-reduce of the (B, V+1)
logits per step, and nothing is read back until the loop ends.

`sample_beam` is the same averaging under the fusion model's beam search (eval_utils.eval_ensemble, eval_utils.py:387-720,
whose bookkeeping is misc/RecurrentFusionModel.py:451-531): all images' beams share one decoder batch of B * beam rows per
member, the candidate / stable-sort / fork / done-beam rules run on the device (rfn_beam_step_topk) on the W best entries of
the averaged log-softmax (rfn_log_softmax_topk), every member's state follows the same fork order.
"""
import torch
import torch.distributed as dist

from . import _native as N
from .fusion_model import _Stepper, _sorted_done_beams


class EnsembleDecoder:
    def __init__(self, models, process_group=None, total_members=None):
        """models: the members held by THIS process (same device, same vocabulary / seq_length).
        process_group: if given, members of all ranks of the group are averaged (total_members = sum over ranks,
        default len(models) * world_size)."""
        if not models:
            raise ValueError('need at least one model')
        self.models = list(models)
        self.group = process_group
        world = dist.get_world_size(process_group) if process_group is not None else 1
        self.n_total = total_members if total_members is not None else len(self.models) * world
        m0 = self.models[0]
        for m in self.models:
            if m.vocab_size != m0.vocab_size or m.seq_length != m0.seq_length:
                raise N.RfnError('ensemble members disagree on vocab_size / seq_length')

    @torch.no_grad()
    def sample(self, fc_feats, att_feats):
        """Greedy ensemble decode -> (seq (B,<=S), seqLogprobs, logprobs_all (B,<=S+1,V+1)) with the reference's
        sample() conventions (finished rows masked to 0, early exit when every row has finished)."""
        m0 = self.models[0]
        B, S, V1 = fc_feats[0].size(0), m0.seq_length, m0.vocab_size + 1
        steppers = []
        for m in self.models:
            comb, h, c, _ = m._prefix(fc_feats, att_feats, False, 0)
            steppers.append(_Stepper(m, comb, h.clone(), c.clone()))
        dev = steppers[0].h.device
        st = N.stream_ptr()
        logit_sum = torch.empty(B, V1, device=dev)
        logit_m = torch.empty(B, V1, device=dev)
        logp_all = torch.empty(B, S + 1, V1, device=dev)
        seq = torch.zeros(B, S, dtype=torch.long, device=dev)
        seq_lp = torch.zeros(B, S, device=dev)
        unf = torch.zeros(S + 1, B, dtype=torch.int32, device=dev)
        it = torch.zeros(B, dtype=torch.long, device=dev)
        for t in range(S + 1):
            if t >= 1:
                prev = logp_all[:, t - 1]
                N.check(N.lib.rfn_greedy_pick(prev.data_ptr(), prev.stride(0), B, V1, t, it.data_ptr(),
                                              seq[:, t - 1].data_ptr(), seq.stride(0), seq_lp[:, t - 1].data_ptr(),
                                              seq_lp.stride(0), unf[t - 1].data_ptr() if t > 1 else None,
                                              unf[t].data_ptr(), st), 'rfn_greedy_pick')
            for j, sp in enumerate(steppers):          # every member embeds the SAME token with its own table
                sp.step(it, out=logit_sum if j == 0 else logit_m, want='logits')
                if j:
                    N.check(N.lib.rfn_axpby_2d(1.0, logit_m.data_ptr(), V1, 1.0, logit_sum.data_ptr(), V1, B, V1, st))
            if self.group is not None:
                dist.all_reduce(logit_sum, op=dist.ReduceOp.SUM, group=self.group)
            N.check(N.lib.rfn_div_2d(logit_sum.data_ptr(), V1, B, V1, float(self.n_total), st))
            out = logp_all[:, t]
            N.check(N.lib.rfn_log_softmax_fwd(logit_sum.data_ptr(), V1, B, V1, B, out.stride(0), 0, out.data_ptr(), st))
        alive = unf[1:].sum(1).tolist()
        t_stop = S + 1
        for t in range(1, S + 1):
            if alive[t - 1] == 0:
                t_stop = t
                break
        return seq[:, :t_stop - 1], seq_lp[:, :t_stop - 1], logp_all[:, :t_stop].contiguous()

    @torch.no_grad()
    def sample_beam(self, fc_feats, att_feats, opt={}):
        """Beam search on the members' averaged distribution -> (seq (B, S), seqLogprobs (B, S), top_seq, top_prob) with
        the conventions of RecurrentFusionModel.sample_beam (best done beam per image; per-image lists of all done beams,
        best first); `self.done_beams` as there.  A one-member ensemble IS that model's sample_beam, bit for bit."""
        W = opt.get('beam_size', 10)
        m0 = self.models[0]
        B, S, V1 = fc_feats[0].size(0), m0.seq_length, m0.vocab_size + 1
        if W > 32 or S > 64 or W > V1:
            raise N.RfnError('beam search supports beam_size <= 32 (and <= V+1) and seq_length <= 64')
        steppers = []
        for m in self.models:
            comb, h, c, _ = m._prefix(fc_feats, att_feats, False, 0)
            steppers.append(_Stepper(m, comb.repeat_interleave(W, dim=1).contiguous(),       # row k * W + q = image k
                                     h.repeat_interleave(W, dim=0).contiguous(), c.repeat_interleave(W, dim=0).contiguous()))
        dev = steppers[0].h.device
        st = N.stream_ptr()
        rows, max_done = B * W, W * S
        bs = torch.zeros(S, B, W, dtype=torch.long, device=dev)
        bl = torch.zeros(S, B, W, device=dev)
        bsum = torch.zeros(B, W, device=dev)
        order = torch.zeros(rows, dtype=torch.int32, device=dev)
        ids = torch.zeros(rows, dtype=torch.long, device=dev)
        done_seq = torch.zeros(B, max_done, S, dtype=torch.long, device=dev)
        done_lp = torch.zeros(B, max_done, S, device=dev)
        done_p = torch.zeros(B, max_done, device=dev)
        done_n = torch.zeros(B, dtype=torch.int32, device=dev)
        active = torch.ones(B, dtype=torch.int32, device=dev)
        topv = torch.empty(rows, W, device=dev)
        topi = torch.empty(rows, W, dtype=torch.int32, device=dev)
        logit_sum = torch.empty(rows, V1, device=dev)
        logit_m = torch.empty(rows, V1, device=dev)
        for t in range(S + 1):
            if t >= 1:
                N.check(N.lib.rfn_beam_step_topk(topv.data_ptr(), topi.data_ptr(), V1, W, S, t, B, max_done, bs.data_ptr(),
                                                 bl.data_ptr(), bsum.data_ptr(), order.data_ptr(), ids.data_ptr(),
                                                 done_seq.data_ptr(), done_lp.data_ptr(), done_p.data_ptr(), done_n.data_ptr(),
                                                 active.data_ptr(), st), 'rfn_beam_step_topk')
                if t == S:
                    break                      # the reference runs one more decoder step whose output is never used
                for sp in steppers:
                    sp.reorder(order)
            for j, sp in enumerate(steppers):
                sp.step(ids, out=logit_sum if j == 0 else logit_m, want='logits')
                if j:
                    N.check(N.lib.rfn_axpby_2d(1.0, logit_m.data_ptr(), V1, 1.0, logit_sum.data_ptr(), V1, rows, V1, st))
            if self.group is not None:
                dist.all_reduce(logit_sum, op=dist.ReduceOp.SUM, group=self.group)
            N.check(N.lib.rfn_div_2d(logit_sum.data_ptr(), V1, rows, V1, float(self.n_total), st))
            N.check(N.lib.rfn_log_softmax_topk(logit_sum.data_ptr(), V1, rows, V1, W, topv.data_ptr(), topi.data_ptr(), st),
                    'rfn_log_softmax_topk')
        seq, seq_lp, top_seq, top_prob, self.done_beams = _sorted_done_beams(done_seq, done_lp, done_p, done_n, S, max_done)
        return seq, seq_lp, top_seq, top_prob

