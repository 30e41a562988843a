"""CPU: the C-ABI library loads, exports every symbol include/rfn.h declares, and its host-side logic
(parameter table, shape validation, workspace queries) behaves -- no kernel is launched."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def native():
    import recurrent_fusion_network_amd._native as N
    return N


def header_symbols():
    src = open(os.path.join(ROOT, 'include', 'rfn.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(rfn_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    N = native()
    syms = header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(N.lib, s), 'librfn_hip.so does not export %s' % s
    # and the binding declares argtypes for every one of them
    assert set(syms) <= set(N.EXPORTS), sorted(set(syms) - set(N.EXPORTS))
    assert N.lib.rfn_abi_version() == N.ABI_VERSION


def test_struct_layouts_match_the_header():
    N = native()
    assert C.sizeof(N.GemmSeg) == 56
    assert C.sizeof(N.GemmProblem) == 32 + 8 * 56
    # ... + gemm_flags (ABI 4) + path_flags, 4 bytes of alignment, probe_events pointer (ABI 6)
    assert C.sizeof(N.Dims) == 8 * 4 + 3 * 8 * 4 + 2 * 4 + 3 * 4 + 4 + 4 + 4 + 8
    assert N.Dims.probe_events.offset == 160 and N.PATH_OPT_PERSIST_ALL == 15
    assert N.GEMM_OPT_LDS_LEAN == 1 and N.GEMM_OPT_NO_DMA == 2             # rfn.h RFN_GEMM_OPT_*


def test_param_table_matches_reference_schema():
    from oracle import rfn_oracle as O
    N = native()
    info = [dict(att_num=196, att_feat_size=2048, fc_feat_size=2048), dict(att_num=64, att_feat_size=1536, fc_feat_size=1536),
            dict(att_num=64, att_feat_size=1280, fc_feat_size=2048), dict(att_num=49, att_feat_size=2208, fc_feat_size=2208),
            dict(att_num=64, att_feat_size=1536, fc_feat_size=1536)]   # the reference's shipped 5 encoders
    cfg = O.make_cfg(info, vocab_size=9487)
    d = N.make_dims(5, 512, 512, 512, 8, 8, 1000, 9488, [f['att_num'] for f in info],
                    [f['att_feat_size'] for f in info], [f['fc_feat_size'] for f in info])
    names = N.param_names(d)
    shapes = O.param_shapes(cfg)
    assert sorted(names) == sorted(shapes) and len(set(names)) == len(names)
    for i, n in enumerate(names):
        r, c = N.param_shape(d, i)
        want = shapes[n]
        assert r == want[0] and r * c == int(__import__('numpy').prod(want)), (n, r, c, want)


def test_bad_configurations_are_rejected_without_launching():
    N = native()
    ok = dict(M=2, R=16, A=16, E=16, T1=3, T2=3, K=20, V1=51, L=[5, 7], D=[24, 40], Fc=[24, 32])
    assert N.lib.rfn_param_count(C.byref(N.make_dims(**ok))) > 0
    for bad in (dict(M=0), dict(R=0), dict(V1=1), dict(T1=0)):
        kw = dict(ok)
        kw.update(bad)
        if kw['M'] == 0:
            kw['L'], kw['D'], kw['Fc'] = [], [], []
        assert N.lib.rfn_param_count(C.byref(N.make_dims(**kw))) == -1
    assert N.lib.rfn_param_count(C.byref(N.make_dims(review_maxout=1, decoder_maxout=1, **ok))) > 0
    d5 = N.make_dims(review_maxout=1, decoder_maxout=1, **ok)
    names5 = N.param_names(d5)
    assert N.param_shape(d5, names5.index('review_steps.0.h2h.weight')) == (5 * 16, 16)
    assert N.param_shape(d5, names5.index('decoder.i2h.bias')) == (5 * 16, 1)
    assert N.param_shape(d5, names5.index('review_steps_individual.0.lstm.0.H2h.weight')) == (4 * 16, 2 * 16)   # fusion_maxout is ignored
    assert N.lib.rfn_param_count(C.byref(N.make_dims(drop_lm=1.0, **ok))) == -1
    assert N.lib.rfn_prefix_ws_bytes(C.byref(N.make_dims(**ok)), 0, 1) == 0
    with pytest.raises(N.RfnError):
        N.check(-4, 'x')
    assert b'workspace' in N.lib.rfn_error_string(-4)


def test_workspace_queries_scale_with_the_batch():
    N = native()
    d = N.make_dims(4, 512, 512, 512, 8, 8, 1000, 9488, [196] * 4, [2048] * 4, [2048] * 4)
    a = N.lib.rfn_prefix_ws_bytes(C.byref(d), 64, 1)
    b = N.lib.rfn_prefix_ws_bytes(C.byref(d), 256, 1)
    inf = N.lib.rfn_prefix_ws_bytes(C.byref(d), 256, 0)
    fixed = 256 << 20            # the split-K scratch (GEMM_WS_FLOATS) does not grow with the batch
    assert 3.5 < (b - fixed) / (a - fixed) <= 4.05 and inf < b
    # the hoisted projections dominate: 4 encoders x (256*196) x (8*512) floats = 3.29 GB
    assert b > 4 * 256 * 196 * 8 * 512 * 4
    assert N.lib.rfn_decoder_ws_bytes(C.byref(d), 256, 17, 1) > N.lib.rfn_decoder_ws_bytes(C.byref(d), 256, 17, 0) > 0
    assert N.lib.rfn_decoder_step_ws_bytes(C.byref(d), 5) > 0


def test_model_shell_schema_and_loud_failure_on_cpu():
    import torch
    import recurrent_fusion_network_amd as R
    from oracle import rfn_oracle as O
    info = [dict(att_num=5, att_feat_size=24, fc_feat_size=24), dict(att_num=7, att_feat_size=40, fc_feat_size=32)]
    cfg = O.make_cfg(info, vocab_size=50, rnn_size=16, input_encoding_size=16, att_hid_size=16, num_review_steps_0=3,
                     num_review_steps=3, top_words_count=20, seq_length=5)
    cfg.caption_model = 'recurrent_fusion_model'
    model = R.setup(cfg)
    shapes = O.param_shapes(cfg)
    sd = model.state_dict()
    assert sorted(sd) == sorted(shapes)
    assert all(tuple(sd[k].shape) == shapes[k] for k in sd)
    model.load_state_dict(O.seeded_params(cfg, 0))           # reference-format checkpoints load
    fc, att, labels, masks, top = O.synthetic_batch(cfg, 3, seed=1)
    with pytest.raises(R._native.RfnError):                   # no CPU fallback
        model(fc, att, labels)
    cfg.caption_model = 'review_net'                          # not provided: only the fusion model and show_tell
    with pytest.raises(Exception, match='not supported'):
        R.setup(cfg)
    cfg.caption_model = 'recurrent_fusion_model'
    cfg.maxout, cfg.review_maxout, cfg.fusion_maxout = 1, 1, 1
    m5 = R.RecurrentFusionModel(cfg)
    shapes5 = O.param_shapes(cfg)
    assert all(tuple(v.shape) == shapes5[k] for k, v in m5.state_dict().items())
    assert m5.decoder.h2h.weight.shape[0] == 5 * 16 and m5.review_steps_individual[0].lstm[0].H2h.weight.shape[0] == 4 * 16
