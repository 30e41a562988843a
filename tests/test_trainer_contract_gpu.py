"""Trainer-facing contract the reference's loops rely on beyond forward/backward:
  * loss.backward(retain_graph=True) called repeatedly on one graph (PPO loop, train_rl.py:190-201);
  * gradients accumulated over two backward passes stay the fused optimizer's operand;
  * optimizer.param_groups[0]['lr'] (utils.set_lr, misc/utils.py:286-290) and optimizer state save / resume
    (optimizer_<id>.pth, train.py:86-88,232-233)."""
import pytest
import torch

from conftest import load_case
from test_model_gpu import build, maxerr, to_dev

pytestmark = pytest.mark.gpu


def _loss(model, crit, fc, att, labels, masks, top):
    lp, reason = model(fc, att, labels)
    return crit(lp, labels[:, 1:], masks[:, 1:], reason, top, 1.0)


@pytest.mark.parametrize('drop', [0.0, 0.3])
def test_backward_twice_over_one_graph(dev, drop):
    import recurrent_fusion_network_amd as R
    cfg, spec, P, batch, gold = load_case('mid')
    cfg.drop_prob_lm = cfg.drop_prob_reason = cfg.drop_prob_fusion = drop
    model = build(cfg, P, dev, train=True)
    fc, att, labels, masks, top = to_dev(batch, dev)
    crit = R.ReviewNetEnsembleCriterion(cfg)
    opt = R.FusedClampAdam(model, lr=0.0)                 # lr 0: the weights stay put between the passes
    torch.manual_seed(11)
    loss = _loss(model, crit, fc, att, labels, masks, top)
    opt.zero_grad()
    loss.backward(retain_graph=True)
    first = {k: p.grad.clone() for k, p in model.named_parameters()}
    for _ in range(2):                                    # ppo_k further passes over the same graph
        opt.zero_grad()
        loss.backward(retain_graph=True)
        for k, p in model.named_parameters():
            assert torch.equal(p.grad, first[k]), k       # same graph, same seed, fixed-order reductions: bit-equal


def test_rl_graph_backward_twice(dev):
    """The PPO loop itself: sample(sample_max=0) graph + reward criterion, backward three times."""
    import recurrent_fusion_network_amd as R
    cfg, spec, P, batch, gold = load_case('mid')
    cfg.use_ppo = 1
    model = build(cfg, P, dev, train=True)
    fc, att, labels, masks, top = to_dev(batch, dev)
    raw = torch.from_numpy(gold['rl_raw_ids'])
    seq, seq_lp, lp_all, reason = model.sample(fc, att, {'sample_max': 0, 'force_ids': raw})
    crit = R.ReviewNetRewardCriterion(cfg)
    reward = torch.from_numpy(gold['rl_reward']).to(dev)
    loss = crit(seq_lp, seq, reward, lp_all, 0.01, reason, top, 1.0, seq_lp.detach(), cfg)
    opt = R.FusedClampAdam(model, lr=0.0)
    grads = []
    for _ in range(3):
        opt.zero_grad()
        loss.backward(retain_graph=True)
        grads.append({k: p.grad.clone() for k, p in model.named_parameters()})
    for k in grads[0]:
        assert torch.equal(grads[0][k], grads[1][k]) and torch.equal(grads[0][k], grads[2][k]), k


def test_accumulated_gradients_reach_the_fused_optimizer(dev):
    import recurrent_fusion_network_amd as R
    cfg, spec, P, batch, gold = load_case('tiny0')
    fc, att, labels, masks, top = to_dev(batch, dev)
    crit = R.ReviewNetEnsembleCriterion(cfg)
    kw = dict(lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5, grad_clip=1e9)
    m1 = build(cfg, P, dev, train=True)
    o1 = R.FusedClampAdam(m1, **kw)
    o1.zero_grad()
    for _ in range(2):                                    # two forward/backward passes, one step
        _loss(m1, crit, fc, att, labels, masks, top).backward()
    for name, flat in m1._last_flat_grads.items():        # .grad and the optimizer operand are the same memory
        p0 = m1.bucket_layout(name)[0][0]
        assert p0.grad.data_ptr() == flat.data_ptr()
    o1.step()
    m2 = build(cfg, P, dev, train=True)
    o2 = R.FusedClampAdam(m2, **kw)
    o2.zero_grad()
    (2.0 * _loss(m2, crit, fc, att, labels, masks, top)).backward()
    g1, g2 = dict(m1.named_parameters()), dict(m2.named_parameters())
    for k in g1:
        assert maxerr(g1[k].grad, g2[k].grad.cpu()) <= 1e-6 + 1e-5 * float(g2[k].grad.abs().max()), k
    o2.step()
    for k in g1:
        sel = (g2[k].grad.abs() > 1e-4).cpu()
        if bool(sel.any()):
            assert float((g1[k].detach().cpu()[sel] - g2[k].detach().cpu()[sel]).abs().max()) < 5e-6, k
    # accumulation over passes and per-bucket all-reduce hooks do not mix: loud error, not a silent partial update
    m1.grad_ready_hook = lambda name, flat: None
    with pytest.raises(R._native.RfnError):
        _loss(m1, crit, fc, att, labels, masks, top).backward()


def test_optimizer_param_groups_and_state_round_trip(dev, tmp_path):
    import recurrent_fusion_network_amd as R
    cfg, spec, P, batch, gold = load_case('tiny1')
    fc, att, labels, masks, top = to_dev(batch, dev)
    crit = R.ReviewNetEnsembleCriterion(cfg)

    def run(model, opt, steps):
        for _ in range(steps):
            opt.zero_grad()
            _loss(model, crit, fc, att, labels, masks, top).backward()
            opt.step()

    kw = dict(lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5, grad_clip=1.0)
    m1 = build(cfg, P, dev, train=True)
    o1 = R.FusedClampAdam(m1, **kw)
    run(m1, o1, 2)
    for group in o1.param_groups:                          # utils.set_lr (misc/utils.py:286-290)
        group['lr'] = 1e-4
    assert o1.lr == 1e-4
    torch.save(o1.state_dict(), str(tmp_path / 'optimizer_x.pth'))
    torch.save(m1.state_dict(), str(tmp_path / 'model_x.pth'))
    run(m1, o1, 2)
    # resume: fresh model + optimizer from the two files, same two further steps
    m2 = build(cfg, torch.load(str(tmp_path / 'model_x.pth')), dev, train=True)
    o2 = R.FusedClampAdam(m2, **kw)
    o2.load_state_dict(torch.load(str(tmp_path / 'optimizer_x.pth')))
    assert o2.step_count == 2 and o2.lr == 1e-4
    run(m2, o2, 2)
    p2 = dict(m2.named_parameters())
    for k, p in m1.named_parameters():
        assert torch.equal(p.detach(), p2[k].detach()), k


def test_optimizer_resumes_from_a_torch_adam_checkpoint(dev, tmp_path):
    """A checkpoint directory written by the reference holds torch.optim.Adam state (train.py:86-88,232-233:
    {'state': {index: step / exp_avg / exp_avg_sq}, 'param_groups'}), indexed by position in model.parameters().
    FusedClampAdam.load_state_dict scatters it into its flat moments; unknown formats fail by name."""
    import recurrent_fusion_network_amd as R
    cfg, spec, P, batch, gold = load_case('tiny0')
    fc, att, labels, masks, top = to_dev(batch, dev)
    crit = R.ReviewNetEnsembleCriterion(cfg)
    kw = dict(lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5)

    def run(model, opt, steps, fused):
        for _ in range(steps):
            opt.zero_grad()
            _loss(model, crit, fc, att, labels, masks, top).backward()
            if not fused:
                R.clip_gradient(opt, 1.0)                     # misc/utils.py:292-296
            opt.step()

    m1 = build(cfg, P, dev, train=True)
    o1 = torch.optim.Adam(m1.parameters(), **kw)             # the reference's optimizer (train.py:69-71)
    run(m1, o1, 2, False)
    torch.save(o1.state_dict(), str(tmp_path / 'optimizer_ref.pth'))
    torch.save(m1.state_dict(), str(tmp_path / 'model_ref.pth'))
    run(m1, o1, 2, False)
    m2 = build(cfg, torch.load(str(tmp_path / 'model_ref.pth')), dev, train=True)
    o2 = R.FusedClampAdam(m2, lr=1.0, grad_clip=1.0)         # hyper-parameters come from the checkpoint
    o2.load_state_dict(torch.load(str(tmp_path / 'optimizer_ref.pth')))
    assert o2.step_count == 2 and o2.lr == 5e-4 and o2.weight_decay == 1e-5
    run(m2, o2, 2, True)
    p2 = dict(m2.named_parameters())
    for k, p in m1.named_parameters():
        # two more Adam steps from the same moments: equal up to the rounding of the two implementations, amplified where
        # |g| is tiny against sqrt(v) (see DESIGN 6 on the first Adam step)
        assert maxerr(p.detach(), p2[k].detach().cpu()) < 2e-5, k
    with pytest.raises(R._native.RfnError):
        o2.load_state_dict({'something': 1})


def test_retained_graph_after_a_weight_update_matches_the_reference_semantics(dev):
    """PPO loop (train_rl.py:190-201): loss.backward(retain_graph=True), optimizer.step(), then backward AGAIN over the
    same graph.  The reference (and the oracle, an ordinary autograd graph) differentiates the activations saved at the
    original forward through the weights as they are now.  model.retain_activations = True gives exactly that; the
    default recomputes the activations at the new weights (self-consistent, documented in fusion_model.py)."""
    import recurrent_fusion_network_amd as R
    from oracle import rfn_oracle as O
    cfg, spec, P, batch, gold = load_case('mid')
    fc, att, labels, masks, top = batch
    lr = 0.05

    # oracle: an in-place update through .data leaves the saved tensors' version alone, as PyTorch 0.3.1 optimizers did
    passes = 4                                             # ppo_k further passes, a weight update before each
    Pg = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
    lp, heads = O.forward(cfg, Pg, fc, att, labels)
    loss = O.xe_criterion(cfg, lp, labels[:, 1:], masks[:, 1:], heads, top, 1.0)
    want = []
    for _ in range(passes):
        loss.backward(retain_graph=True)
        want.append({k: v.grad for k, v in Pg.items()})
        for v in Pg.values():
            v.data.add_(v.grad, alpha=-lr)
            v.grad = None

    def run_passes(retain):
        model = build(cfg, P, dev, train=True)
        model.retain_activations = retain
        crit = R.ReviewNetEnsembleCriterion(cfg)
        l_ = _loss(model, crit, *to_dev(batch, dev))
        got = []
        for _ in range(passes):
            l_.backward(retain_graph=True)
            got.append({k: p.grad.clone() for k, p in model.named_parameters()})
            with torch.no_grad():
                for p in model.parameters():
                    p.data.add_(p.grad, alpha=-lr)
                    p.grad = None
            model._last_flat_grads.clear()
        return got

    got = run_passes(True)
    for n in range(passes):
        for k, w in want[n].items():
            assert maxerr(got[n][k], w) <= 1e-5 + 1e-3 * float(w.abs().max()), (n, k)
        # logit.weight's gradient is dlogits^T h: activations and saved log-probs only -> unchanged by the updates, bit for bit
        assert torch.equal(got[n]['logit.weight'], got[0]['logit.weight'])
    recomputed = run_passes(False)
    assert not torch.equal(recomputed[1]['logit.weight'], recomputed[0]['logit.weight'])
    worst = max(float((recomputed[1][k].cpu() - w).abs().max()) / (1e-5 + 1e-3 * float(w.abs().max())) for k, w in want[1].items())
    assert worst > 1.0          # the recompute path is a different (self-consistent) gradient: the flag matters


def test_graphed_train_step_is_bit_identical_to_the_eager_step(dev):
    """graphed.GraphedTrainStep: zero_grad + forward + criterion + backward + clamp + Adam captured once in a HIP graph and
    replayed on three different batches = the same three eager steps, bit for bit (loss, every parameter, both Adam moments,
    `.grad`), including Adam's step-dependent bias correction, which the replay reads from device memory
    (rfn_adam_step_multi_coef).  Constructing the wrapper must not train; dropout > 0 is refused."""
    import recurrent_fusion_network_amd as R
    from recurrent_fusion_network_amd.graphed import GraphedTrainStep
    cfg, spec, P, batch, gold = load_case('mid')
    fc, att, labels, masks, top = to_dev(batch, dev)
    B = labels.size(0)
    perms = [torch.arange(B, device=dev), torch.arange(B, device=dev).flip(0), torch.roll(torch.arange(B, device=dev), 2)]
    batches = [([f[p] for f in fc], [a[p] for a in att], labels[p], masks[p], top[p]) for p in perms]
    kw = dict(lr=5e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5, grad_clip=0.01)      # a clamp that bites, a big lr
    crit = R.ReviewNetEnsembleCriterion(cfg)

    eager = build(cfg, P, dev, train=True)
    o1 = R.FusedClampAdam(eager, **kw)
    want = []
    for b in batches:
        o1.zero_grad()
        lp, reason = eager(b[0], b[1], b[2])
        loss = crit(lp, b[2][:, 1:], b[3][:, 1:], reason, b[4], 1.0)
        loss.backward()
        o1.step()
        want.append(loss.detach().clone())

    model = build(cfg, P, dev, train=True)
    o2 = R.FusedClampAdam(model, **kw)
    before = {k: p.detach().clone() for k, p in model.named_parameters()}
    g = GraphedTrainStep(model, crit, o2, *batches[0])
    assert o2.step_count == 0 and all(torch.equal(p, before[k]) for k, p in model.named_parameters())   # capture did not train
    for b, w in zip(batches, want):
        loss = g(*b)
        assert torch.equal(loss.detach(), w)
    assert o2.step_count == 3
    for (k, p), (_, q) in zip(model.named_parameters(), eager.named_parameters()):
        assert torch.equal(p, q), k
        assert torch.equal(p.grad, q.grad), k
    for name in o1.flat:            # the moments of every parameter (the 16-B padding between parameters holds no state)
        params, offs, _ = model.bucket_layout(name)
        for p_, o in zip(params, offs):
            for k in ('m', 'v'):
                assert torch.equal(o1.flat[name][k][o:o + p_.numel()], o2.flat[name][k][o:o + p_.numel()]), (name, k)
    # what a capture would freeze is refused
    cfg.drop_prob_lm = 0.3
    drop = build(cfg, P, dev, train=True)
    with pytest.raises(R._native.RfnError):
        GraphedTrainStep(drop, crit, R.FusedClampAdam(drop, **kw), *batches[0])


def test_graphed_train_step_serves_batches_of_any_caption_length_and_refuses_what_it_cannot_replay(dev):
    """the capture used to freeze the example batch's decoder-step count and replay it on whatever labels were
    copied in.  Now the step is captured with all seq_length + 1 decoder steps: a batch whose longest caption is shorter
    replays to the eager result (the masks add exact zeros on the extra steps), a batch whose captions are LONGER than the
    example's replays correctly too, and shape / hyper-parameter / mode changes are refused instead of silently ignored."""
    import recurrent_fusion_network_amd as R
    from recurrent_fusion_network_amd.graphed import GraphedTrainStep
    cfg, spec, P, batch, gold = load_case('mid')
    fc, att, labels, masks, top = to_dev(batch, dev)
    ncol = labels.size(1)

    def cut(keep):      # captions of at most `keep` words: END at column keep + 1, zeros and zero masks behind it
        lab, msk = labels.clone(), masks.clone()
        lab[:, keep + 1:] = 0
        msk[:, keep + 2:] = 0
        return (fc, att, lab, msk, top)

    short, longer = cut(3), cut(ncol - 3)
    kw = dict(lr=5e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5, grad_clip=0.01)
    crit = R.ReviewNetEnsembleCriterion(cfg)

    def eager_steps(bs):
        m = build(cfg, P, dev, train=True)
        o = R.FusedClampAdam(m, **kw)
        losses = []
        for b in bs:
            o.zero_grad()
            lp, reason = m(b[0], b[1], b[2])
            assert lp.size(1) == m._decoder_steps(b[2])
            loss = crit(lp, b[2][:, 1:], b[3][:, 1:], reason, b[4], 1.0)
            loss.backward()
            o.step()
            losses.append(float(loss))
        return m, losses

    eager, want = eager_steps([short, longer, short])
    assert eager._decoder_steps(short[2]) == 4 and eager._decoder_steps(longer[2]) == ncol - 2
    model = build(cfg, P, dev, train=True)
    opt = R.FusedClampAdam(model, **kw)
    g = GraphedTrainStep(model, crit, opt, *short)          # captured on the SHORT batch
    assert g.steps == ncol - 1 and model.fixed_decoder_steps is None
    got = [float(g(*b)) for b in (short, longer, short)]
    for a, b in zip(got, want):
        assert abs(a - b) <= 1e-5 * max(1.0, abs(b)), (got, want)
    for (k, p), (_, q) in zip(model.named_parameters(), eager.named_parameters()):
        assert float((p - q).abs().max()) <= 1e-6 + 1e-4 * float(q.abs().max()), k

    # a capture at the example's own count refuses a batch that needs another count (it used to drop the extra tokens)
    m2 = build(cfg, P, dev, train=True)
    g2 = GraphedTrainStep(m2, crit, R.FusedClampAdam(m2, **kw), *short, full_length=False)
    assert g2.steps == 4
    g2(*short)
    with pytest.raises(R._native.RfnError, match='decoder steps'):
        g2(*longer)
    # shapes are checked before the copy into the static buffers (copy_ would broadcast a one-row batch)
    with pytest.raises(R._native.RfnError, match='captured step holds'):
        g(fc, att, short[2][:1], short[3], top)
    with pytest.raises(R._native.RfnError, match='captured step holds'):
        g([f[:1] for f in fc], att, short[2], short[3], top)
    # what the captured launches hold as arguments must not change silently
    opt.param_groups[0]['weight_decay'] = 0.0
    with pytest.raises(R._native.RfnError, match='weight_decay'):
        g(*short)
    opt.param_groups[0]['weight_decay'] = kw['weight_decay']
    model.ss_prob = 0.25
    with pytest.raises(R._native.RfnError):
        g(*short)
    model.ss_prob = 0.0
    model.eval()
    with pytest.raises(R._native.RfnError, match='training'):
        g(*short)
    model.train()
    opt.set_lr(1e-3)            # the one thing a replay re-reads
    g(*short)


def test_overlapped_update_is_bit_identical_to_the_single_launch(dev):
    """parallel.OverlappedUpdate (opt-in, one process): every bucket's clamp + Adam on a side stream as soon as backward has
    finished the bucket, the next forward waiting for it where it first reads parameters -- three steps give the same loss,
    parameters and moments as the one launch after backward, bit for bit (measured slower on MI355X: profiles/r05_notes.md;
    it stays an A/B hook)."""
    import recurrent_fusion_network_amd as R
    from recurrent_fusion_network_amd import parallel as DP
    cfg, spec, P, batch, gold = load_case('mid')
    fc, att, labels, masks, top = to_dev(batch, dev)
    kw = dict(lr=5e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5, grad_clip=0.01)
    crit = R.ReviewNetEnsembleCriterion(cfg)
    runs = []
    for overlapped in (False, True):
        model = build(cfg, P, dev, train=True)
        opt = R.FusedClampAdam(model, **kw)
        if overlapped:
            DP.OverlappedUpdate(model, opt)
        losses = []
        for _ in range(3):
            opt.zero_grad()
            lp, reason = model(fc, att, labels)
            loss = crit(lp, labels[:, 1:], masks[:, 1:], reason, top, 1.0)
            loss.backward()
            opt.step()
            losses.append(loss.detach().clone())
        torch.cuda.synchronize()
        assert opt.step_count == 3
        runs.append((model, opt, losses))
    (m0, o0, l0), (m1, o1, l1) = runs
    for a, b in zip(l0, l1):
        assert torch.equal(a, b)
    for (k, p), (_, q) in zip(m0.named_parameters(), m1.named_parameters()):
        assert torch.equal(p, q), k
    for name in o0.flat:            # the moments of every parameter (the 16-B padding between parameters holds no state)
        params, offs, _ = m0.bucket_layout(name)
        for p_, o in zip(params, offs):
            for k in ('m', 'v'):
                assert torch.equal(o0.flat[name][k][o:o + p_.numel()], o1.flat[name][k][o:o + p_.numel()]), (name, k)
