"""ShowTellModel (BASELINE config 1: single-encoder plumbing check on the CPU; SURVEY.md section 2 "reuse stock
nn.LSTM, no kernel work").

NOT on the accelerated path: this is the reference's simplest captioner (misc/ShowTellModel.py:10-240) restated on
stock PyTorch modules so that `models.setup(opt)` serves `caption_model = 'show_tell'` and the trainer / eval
plumbing (forward -> LanguageModelCriterion, greedy sample) can be exercised without a GPU.  Same constructor fields,
same `state_dict` keys (`img_embed`, `core` = nn.LSTM(bias=False), `embed`, `logit`) and the same step conventions:
step 0 feeds the image embedding, step 1 the BOS token 0, outputs start at step 1, the loop stops at the first
all-zero label column (>= 2).  `sample_beam` restates the reference's per-image search (:95-185) with its own rules, which
differ from the fusion model's: a beam that has emitted END keeps competing (no skip), every beam alive at the last step is
recorded as done.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class ShowTellModel(nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.vocab_size = opt.vocab_size
        self.input_encoding_size = opt.input_encoding_size
        self.rnn_type = getattr(opt, 'rnn_type', 'lstm')
        self.rnn_size = opt.rnn_size
        self.num_layers = getattr(opt, 'num_layers', 1)
        self.drop_prob_lm = opt.drop_prob_lm
        self.seq_length = opt.seq_length
        self.fc_feat_size = opt.fc_feat_size
        self.use_cuda = getattr(opt, 'use_cuda', 0)
        self.ss_prob = 0.0
        if self.rnn_type.lower() != 'lstm':
            raise ValueError('show_tell: only rnn_type lstm is provided')
        self.img_embed = nn.Linear(self.fc_feat_size, self.input_encoding_size)
        self.core = nn.LSTM(self.input_encoding_size, self.rnn_size, self.num_layers, bias=False,
                            dropout=self.drop_prob_lm if self.num_layers > 1 else 0.0)
        self.embed = nn.Embedding(self.vocab_size + 1, self.input_encoding_size)
        self.logit = nn.Linear(self.rnn_size, self.vocab_size + 1)
        self.init_weights()

    def init_weights(self):                       # misc/ShowTellModel.py:31-35
        self.embed.weight.data.uniform_(-0.1, 0.1)
        self.logit.weight.data.uniform_(-0.1, 0.1)
        self.logit.bias.data.zero_()

    def _zero_state(self, like, n):
        z = like.new_zeros(self.num_layers, n, self.rnn_size)
        return (z, z.clone())

    def _advance(self, xt, state):
        """one LSTM step + log-softmax over the vocabulary"""
        out, state = self.core(xt.unsqueeze(0), state)
        return F.log_softmax(self.logit(out.squeeze(0)), dim=1), state

    def forward(self, fc_feats, att_feats, seq):
        """-> log_prob (B, T, V+1): outputs of steps 1.. (the image step's output is dropped), :47-81."""
        n = fc_feats.size(0)
        state = self._zero_state(fc_feats, n)
        _, state = self._advance(self.img_embed(fc_feats), state)
        steps, prev = [], None
        for i in range(1, seq.size(1)):
            tok = seq[:, i - 1]
            if i >= 2 and bool((tok == 0).all()):          # every caption has ended (:72-73)
                break
            if i >= 2 and self.ss_prob > 0.0:              # scheduled sampling from the previous distribution (:57-68)
                coin = torch.rand(n, device=fc_feats.device) < self.ss_prob
                if bool(coin.any()):
                    draw = torch.multinomial(torch.exp(prev.detach()), 1).view(-1)
                    tok = torch.where(coin, draw, tok)
            prev, state = self._advance(self.embed(tok), state)
            steps.append(prev)
        return torch.stack(steps, 1).contiguous()

    def sample(self, fc_feats, att_feats, opt={}):
        """-> (seq (B, <=S), seqLogprobs, logprobs_all (B, <=S+1, V+1)); greedy or temperature multinomial (:186-240)."""
        if opt.get('beam_size', 1) > 1:
            return self.sample_beam(fc_feats, att_feats, opt)
        sample_max, temperature = opt.get('sample_max', 1), opt.get('temperature', 1.0)
        n = fc_feats.size(0)
        state = self._zero_state(fc_feats, n)
        _, state = self._advance(self.img_embed(fc_feats), state)
        it = torch.zeros(n, dtype=torch.long, device=fc_feats.device)          # BOS
        seq, seq_lp, all_lp, alive = [], [], [], None
        for t in range(1, self.seq_length + 2):
            if t >= 2:
                if sample_max:
                    picked, it = torch.max(logp.detach(), 1)
                else:
                    p = torch.exp(logp.detach() if temperature == 1.0 else logp.detach() / temperature)
                    it = torch.multinomial(p, 1).view(-1)
                    picked = logp.gather(1, it.view(-1, 1)).view(-1)
                alive = (it > 0) if alive is None else alive & (it > 0)
                if not bool(alive.any()):
                    break
                seq.append(it * alive.long())
                seq_lp.append(picked)
            logp, state = self._advance(self.embed(it), state)
            all_lp.append(logp)
        if not seq:
            e = fc_feats.new_zeros(n, 0)
            return e.long(), e, torch.stack(all_lp, 1)
        return torch.stack(seq, 1), torch.stack(seq_lp, 1), torch.stack(all_lp, 1).contiguous()


    def sample_beam(self, fc_feats, att_feats, opt={}):
        """misc/ShowTellModel.py:95-185 -> (seq (B, S) int64, seqLogprobs (B, S)); side effect `self.done_beams[k]` = the
        finished beams of image k, best first, as dicts {'seq', 'logps', 'p'}.

        One image at a time on `beam_size` rows, as the reference does.  Step 0 feeds the image embedding, step 1 BOS; from
        step 2 on the candidates are (sorted-vocabulary column c, beam q) in that nesting order -- only beam 0 at step 2 --
        scored by the beam's running sum + the column's log-prob in float32, ordered by a STABLE sort on -p; the best
        `beam_size` fork their prefix and LSTM state; a beam whose new token is END (0), and every beam at the last step,
        is appended to the done list.  Unlike the fusion model's search (:475 there) a beam that already ended is not
        skipped: it feeds token 0 and keeps competing -- kept, because it changes which beams survive."""
        beam_size = opt.get('beam_size', 10)
        n, S = fc_feats.size(0), self.seq_length
        assert beam_size <= self.vocab_size + 1, 'lets assume this for now'
        seq = torch.zeros(S, n, dtype=torch.long)
        seq_lp = torch.zeros(S, n)
        self.done_beams = [[] for _ in range(n)]
        with torch.no_grad():
            for k in range(n):
                state = self._zero_state(fc_feats, beam_size)
                beam_seq = torch.zeros(S, beam_size, dtype=torch.long)
                beam_lp = torch.zeros(S, beam_size)
                beam_sum = torch.zeros(beam_size)
                logp = None
                for t in range(S + 2):
                    if t == 0:
                        xt = self.img_embed(fc_feats[k:k + 1]).expand(beam_size, self.input_encoding_size)
                    elif t == 1:
                        xt = self.embed(torch.zeros(beam_size, dtype=torch.long, device=fc_feats.device))
                    else:
                        ys, ix = torch.sort(logp.float().cpu(), 1, True)
                        rows = 1 if t == 2 else beam_size
                        cols = min(beam_size, ys.size(1))
                        total = beam_sum[:rows, None] + ys[:rows, :cols]                  # float32, as the reference sums
                        cands = [(c, q) for c in range(cols) for q in range(rows)]        # column-major candidate order
                        cands.sort(key=lambda cq: -float(total[cq[1], cq[0]]))            # stable
                        prev_seq, prev_lp = beam_seq[:t - 2].clone(), beam_lp[:t - 2].clone()
                        new_state = [x.clone() for x in state]
                        for vix in range(beam_size):
                            c, q = cands[vix]
                            if t > 2:
                                beam_seq[:t - 2, vix] = prev_seq[:, q]
                                beam_lp[:t - 2, vix] = prev_lp[:, q]
                            for dst, src in zip(new_state, state):
                                dst[:, vix] = src[:, q]
                            beam_seq[t - 2, vix] = ix[q, c]
                            beam_lp[t - 2, vix] = ys[q, c]
                            beam_sum[vix] = total[q, c]
                            if int(ix[q, c]) == 0 or t == S + 1:
                                self.done_beams[k].append({'seq': beam_seq[:, vix].clone(), 'logps': beam_lp[:, vix].clone(),
                                                           'p': float(beam_sum[vix])})
                        state = tuple(new_state)
                        xt = self.embed(beam_seq[t - 2].to(fc_feats.device))
                    logp, state = self._advance(xt, state)
                self.done_beams[k].sort(key=lambda b: -b['p'])                            # stable, best first (:181)
                seq[:, k] = self.done_beams[k][0]['seq']
                seq_lp[:, k] = self.done_beams[k][0]['logps']
        return seq.t(), seq_lp.t()


class LanguageModelCriterion(nn.Module):
    """misc/utils.py:252-282: masked NLL (optionally label-smoothed) divided by the batch size; plain torch (CPU
    plumbing for show_tell -- the fusion path uses ReviewNetEnsembleCriterion's HIP kernels)."""

    def __init__(self, opt):
        super().__init__()
        self.use_label_smoothing = opt.use_label_smoothing
        self.label_smoothing_epsilon = opt.label_smoothing_epsilon

    def forward(self, inp, target, mask):
        n, T, K = inp.shape
        target, mask = target[:, :T], mask[:, :T].to(inp.dtype)
        nll = -inp.gather(2, target.unsqueeze(2)).squeeze(2)
        if self.use_label_smoothing:
            eps = self.label_smoothing_epsilon
            nll = (1.0 - eps) * nll - eps / K * inp.sum(2)
        return (nll * mask).sum() / n
