"""ShowTellModel (BASELINE config 1: single-encoder plumbing check on the CPU; SURVEY.md section 2 "reuse stock
nn.LSTM, no kernel work").

NOT on the accelerated path: this is the reference's simplest captioner (misc/ShowTellModel.py:10-240) restated on
stock PyTorch modules so that `models.setup(opt)` serves `caption_model = 'show_tell'` and the trainer / eval
plumbing (forward -> LanguageModelCriterion, greedy sample) can be exercised without a GPU.  Same constructor fields,
same `state_dict` keys (`img_embed`, `core` = nn.LSTM(bias=False), `embed`, `logit`) and the same step conventions:
step 0 feeds the image embedding, step 1 the BOS token 0, outputs start at step 1, the loop stops at the first
all-zero label column (>= 2).  Beam search is not restated (the fusion model's device-resident beam is the product's).
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
            raise NotImplementedError('show_tell: beam search is not provided (see the module docstring)')
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
