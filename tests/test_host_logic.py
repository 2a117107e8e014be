"""CPU tests of the host-side (integer / layout / argument) logic that surrounds the HIP kernels; no kernel runs."""
import argparse

import numpy as np
import pytest
import torch


def test_get_subsample_follows_reference_rules():
    """nets_utils.py:390-468: transformer -> [1]; vgg* never subsamples in the RNN layers; *p types read the string"""
    from espnet_amd.nets.e2e_asr import get_subsample
    ns = argparse.Namespace(elayers=3, etype="blstmp", subsample="1_2_2_1_1")
    assert get_subsample(ns, "asr", "rnn").tolist() == [1, 2, 2, 1]
    ns.etype = "vggblstmp"
    assert get_subsample(ns, "asr", "rnn").tolist() == [1, 1, 1, 1]
    ns.etype = "blstm"
    assert get_subsample(ns, "asr", "rnn-t").tolist() == [1, 1, 1, 1]
    assert get_subsample(ns, "asr", "transformer").tolist() == [1]
    with pytest.raises(ValueError):
        get_subsample(ns, "mt", "rnn")


def test_prepare_loss_inputs_matches_reference_semantics(oracle):
    """transducer/utils.py:9-53: blank-prefixed decoder input, int32 targets padded with blank, lengths from a mask"""
    from espnet_amd.nets.transducer.utils import prepare_loss_inputs
    ys = torch.tensor([[3, 4, 5, -1], [2, -1, -1, -1], [1, 2, 3, 4]])
    mask = torch.tensor([[[1, 1, 1, 1, 0]], [[1, 1, 0, 0, 0]], [[1, 1, 1, 1, 1]]], dtype=torch.bool)
    ys_in, target, pred_len, target_len = prepare_loss_inputs(ys, mask)
    assert ys_in.tolist() == [[0, 3, 4, 5, 0], [0, 2, 0, 0, 0], [0, 1, 2, 3, 4]]
    assert target.dtype == torch.int32 and target.tolist() == [[3, 4, 5, 0], [2, 0, 0, 0], [1, 2, 3, 4]]
    assert pred_len.tolist() == [4, 2, 5] and target_len.tolist() == [3, 1, 4]
    o_in, o_tgt, o_len = oracle.rnnt_prepare(ys)
    assert torch.equal(o_in, ys_in) and torch.equal(o_tgt, target) and torch.equal(o_len, target_len)
    _, _, pl, _ = prepare_loss_inputs(ys, [7, 6, 5])
    assert pl.tolist() == [7, 6, 5]


def test_block_arch_validation():
    """transducer/blocks.py:39-222: malformed block lists are rejected with the reference's conditions"""
    from espnet_amd.nets.transducer.blocks import check_and_prepare
    conf = dict(type="conformer", d_hidden=64, d_ff=96, heads=4, macaron_style=True, use_conv_mod=True, conv_mod_kernel=7)
    layer, odim, dr, pdr, out = check_and_prepare("encoder", [conf, dict(conf, **{"dropout-rate": 0.1})], "conv2d")
    assert (layer, odim, out) == ("conformer-conv2d", 64, 64) and dr == 0.1 and pdr == 0.0
    with pytest.raises(ValueError):
        check_and_prepare("encoder", [{k: v for k, v in conf.items() if k != "conv_mod_kernel"}], "conv2d")
    with pytest.raises(ValueError):
        check_and_prepare("encoder", [conf, dict(conf, d_hidden=32)], "conv2d")
    with pytest.raises(NotImplementedError):
        check_and_prepare("encoder", [conf, dict(type="transformer", d_hidden=64, d_ff=96, heads=4)], "conv2d")
    with pytest.raises(ValueError):
        check_and_prepare("encoder", [conf, dict(d_hidden=64)], "conv2d")          # second block without a type


def test_flat_arena_groups_qkv_projections():
    """FlatParams lays linear_q / linear_k / linear_v weights (and biases) back to back: the precondition of the
    fused [3D, D] projection; every parameter still aliases its slice of the arena"""
    from espnet_amd import train
    from espnet_amd.nets.modules import MultiHeadedAttention, PositionwiseFeedForward
    torch.manual_seed(0)
    model = torch.nn.ModuleDict(dict(a=MultiHeadedAttention(4, 32, 0.0), f=PositionwiseFeedForward(32, 48, 0.0),
                                     b=MultiHeadedAttention(2, 16, 0.0)))
    before = {k: v.detach().clone() for k, v in model.state_dict().items()}
    flat = train.FlatParams(model)
    for k, v in model.state_dict().items():
        assert torch.equal(v, before[k])                         # values survive the move into the arena
    for att in (model["a"], model["b"]):
        q, k, v = att.linear_q.weight, att.linear_k.weight, att.linear_v.weight
        assert k.data_ptr() == q.data_ptr() + q.numel() * 4 and v.data_ptr() == k.data_ptr() + k.numel() * 4
        bq, bk, bv = att.linear_q.bias, att.linear_k.bias, att.linear_v.bias
        assert bk.data_ptr() == bq.data_ptr() + bq.numel() * 4 and bv.data_ptr() == bk.data_ptr() + bk.numel() * 4
        gq, gk = q._eamd_grad, k._eamd_grad
        assert gk.data_ptr() == gq.data_ptr() + gq.numel() * 4
    assert all(o % 8 == 0 for o in flat.offsets) and len(set(map(id, flat.params))) == len(list(model.parameters()))
    assert sorted(flat.offsets) == flat.offsets                  # params list is in layout order (bucket builder relies on it)


def test_flat_arena_groups_stack_wide_projections():
    """FlatParams lays linear_k / linear_v of ALL source-attention modules of the decoder stack and linear_pos of ALL
    encoder layers back to back (one GEMM each, functional.SharedProjFn); self-attention q / k / v stay grouped per module;
    state_dict names and values are untouched; the phased data-parallel plan still gets contiguous arena ranges"""
    import argparse
    from espnet_amd import train
    from espnet_amd.nets.e2e_asr_conformer import E2E
    ns = argparse.Namespace(
        adim=64, aheads=2, elayers=4, eunits=48, dlayers=3, dunits=48, mtlalpha=0.3, lsm_weight=0.1, dropout_rate=0.0,
        transformer_length_normalized_loss=False, transformer_encoder_pos_enc_layer_type="rel_pos",
        transformer_encoder_selfattn_layer_type="rel_selfattn", macaron_style=True, use_cnn_module=True, cnn_module_kernel=7)
    torch.manual_seed(1)
    model = E2E(20, 30, ns)
    before = {k: v.detach().clone() for k, v in model.state_dict().items()}
    flat = train.FlatParams(model)
    assert list(model.state_dict().keys()) == list(before.keys())
    for k, v in model.state_dict().items():
        assert torch.equal(v, before[k])
    def run(ts):
        return all(b.data_ptr() == a.data_ptr() + a.numel() * 4 for a, b in zip(ts, ts[1:]))
    dec = model.decoder.decoders
    kv_w = [w for m in dec for w in (m.src_attn.linear_k.weight, m.src_attn.linear_v.weight)]
    kv_b = [w for m in dec for w in (m.src_attn.linear_k.bias, m.src_attn.linear_v.bias)]
    assert run(kv_w) and run(kv_b) and run([w._eamd_grad for w in kv_w]) and run([w._eamd_grad for w in kv_b])
    pos = [m.self_attn.linear_pos.weight for m in model.encoder.encoders]
    assert run(pos) and run([w._eamd_grad for w in pos])
    for m in list(model.encoder.encoders) + [d.self_attn for d in dec]:
        att = m.self_attn if hasattr(m, "self_attn") else m
        assert run([att.linear_q.weight, att.linear_k.weight, att.linear_v.weight])
    assert sorted(flat.offsets) == flat.offsets and len(set(map(id, flat.params))) == len(list(model.parameters()))
    # arena ranges of the phased backward (decoder / upper encoder layers / ... / input layer + lowest layers + linear_pos group)
    step = train.GraphedDataParallelStep.__new__(train.GraphedDataParallelStep)
    step.model, step.flat, step.ranges, step._stack = model, flat, [(0, flat.numel)], None
    step._plan_phases()
    assert len(step.ranges) >= 3 and step.ranges[0][0] == 0 or min(r[0] for r in step.ranges) == 0
    cover = sorted(step.ranges)
    assert cover[0][0] == 0 and cover[-1][1] == flat.numel and all(a[1] == b[0] for a, b in zip(cover, cover[1:]))


def test_bucketed_graph_step_bucket_key_from_host_lengths():
    """BucketedGraphStep.bucket rounds (T, L) up to the bucket edges and takes the label lengths from the host when given"""
    from espnet_amd import train
    class M:  # noqa: D401
        ignore_id = -1
    st = train.BucketedGraphStep.__new__(train.BucketedGraphStep)
    st.model, st.t_edge, st.l_edge = M(), 64, 8
    xs = torch.zeros(3, 700, 4)
    ys = torch.full((3, 20), -1, dtype=torch.long)
    ys[0, :13] = 5
    assert st.bucket(xs, [700, 650, 600], ys) == (3, 704, 16)
    assert st.bucket(xs, torch.tensor([640, 600, 500]), ys, olens=[13, 4, 2]) == (3, 640, 16)
    assert st.bucket(xs, [65, 3, 2], ys, olens=torch.tensor([17, 1, 1])) == (3, 128, 24)


def test_unsupported_variants_fail_loudly():
    from espnet_amd.nets.rnn.attentions import initial_att
    from espnet_amd.nets.rnn.encoders import Encoder
    with pytest.raises(ValueError):
        initial_att("no_such_attention", 8, 8, 2, 4, 3, 2, 1)
    with pytest.raises(ValueError):
        Encoder("brnnp", 10, 1, 4, 4, np.ones(2, dtype=np.int64), 0.0)
    # scheduled sampling (sampling_probability > 0) is implemented since round 3: tests/test_gpu_rnn.py covers it


def test_beam_step_log_replay_on_the_host():
    """_BatchLog: the host side of a device-resident beam search - rows of the step log [step, score, token, per-scorer
    scores, prefix] are fetched every sync_every steps and the reference's bookkeeping is replayed from them
    (beam_search.py:404-458): a slot whose token is <eos> (or that reaches its utterance's length cap) becomes an ended
    hypothesis with its prefix and scores, empty slots (-inf) are skipped, an utterance stops when nothing is alive or
    end_detect says so, and the search is over when every utterance has stopped."""
    from espnet_amd.nets.beam_search import _BatchLog

    class Scorer:
        def final_score(self, state):
            return 0.25

    class BS:
        beam_size, eos, sync_every, apply_final_score = 2, 9, 2, True
        full_scorers, part_scorers, weights = {"dec": Scorer()}, {}, {"dec": 0.5}

    allk, W = ["dec"], 6
    ninf = -float("inf")

    def row(step, slots):                      # slots: (score, token, scorer score, prefix list)
        r = torch.zeros(len(slots), 3 + len(allk) + W)
        for s, (sc, tok, ds, pre) in enumerate(slots):
            r[s, 0], r[s, 1], r[s, 2], r[s, 3] = step, sc, tok, ds
            r[s, 4: 4 + len(pre)] = torch.tensor(pre, dtype=torch.float32)
        return r

    log = _BatchLog(BS(), 2, [3, 4], 0.5, allk)                     # two utterances, length caps 3 and 4
    # step 0: utterance 0 slot 0 ends with <eos>; utterance 1: one live slot, one empty
    assert not log.add(row(0, [(-1.0, 9, -0.5, [9, 9]), (-2.0, 3, -1.0, [9, 3]), (-1.5, 4, -0.7, [9, 4]), (ninf, 0, 0.0, [9, 0])]), last=False)
    assert log.pending and not log.ended[0]                          # nothing fetched before sync_every rows are in
    # step 1: utterance 0 alive; utterance 1 ends its only live slot
    assert not log.add(row(1, [(-2.5, 5, -1.2, [9, 3, 5]), (ninf, 0, 0.0, [9, 0, 0]), (-2.0, 9, -0.9, [9, 4, 9]), (ninf, 0, 0.0, [9, 0, 0])]),
                       last=False)
    assert not log.pending
    assert [h.yseq.tolist() for h in log.ended[0]] == [[9, 9]] and [h.yseq.tolist() for h in log.ended[1]] == [[9, 4, 9]]
    e0 = log.ended[0][0]
    assert abs(float(e0.score) - (-1.0 + 0.5 * 0.25)) < 1e-6 and abs(e0.scores["dec"] - (-0.5 + 0.25)) < 1e-6      # final scores added
    assert log.stopped == [False, True]                              # utterance 1 has nothing alive any more
    # "full" mode with the candidate-selection kernels: a winner at log-zero level tells the host that a token outside the
    # candidates could have won - the search is to be repeated on the tensor expressions (beam_search._OutsideCandidates)
    from espnet_amd.nets.beam_search import _OutsideCandidates

    class BSFull(BS):
        partial_mode, candidate_select = "full", True
    log2 = _BatchLog(BSFull(), 2, [3, 4], 0.5, allk)
    log2.add(row(0, [(-1.0, 3, -0.5, [9, 3]), (-3.1e9, 4, -1.0, [9, 4]), (-1.5, 4, -0.7, [9, 4]), (ninf, 0, 0.0, [9, 0])]), last=False)
    with pytest.raises(_OutsideCandidates):
        log2.add(row(1, [(-2.5, 5, -1.2, [9, 3, 5]), (ninf, 0, 0.0, [9, 0, 0]), (-2.0, 9, -0.9, [9, 4, 9]), (ninf, 0, 0.0, [9, 0, 0])]), last=False)
    # step 2 = the length cap of utterance 0 (maxlen 3): its live slot ends with <eos> appended
    assert log.add(row(2, [(-3.0, 6, -1.5, [9, 3, 5, 6]), (ninf, 0, 0.0, [9, 0, 0, 0]), (ninf, 0, 0.0, [9, 0, 0, 0]), (ninf, 0, 0.0, [9, 0, 0, 0])]),
                   last=True)
    assert [h.yseq.tolist() for h in log.ended[0]] == [[9, 9], [9, 3, 5, 6, 9]] and log.stopped == [True, True]
    res = log.results()
    assert [h.yseq.tolist() for h in res[0]] == [[9, 9], [9, 3, 5, 6, 9]] and len(res[1]) == 1                    # best first
