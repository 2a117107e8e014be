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


def test_unsupported_variants_fail_loudly():
    from espnet_amd.nets.rnn.attentions import initial_att
    from espnet_amd.nets.rnn.encoders import Encoder
    with pytest.raises(ValueError):
        initial_att("no_such_attention", 8, 8, 2, 4, 3, 2, 1)
    with pytest.raises(ValueError):
        Encoder("brnnp", 10, 1, 4, 4, np.ones(2, dtype=np.int64), 0.0)
    from espnet_amd.nets.rnn.decoders import Decoder
    with pytest.raises(NotImplementedError):
        Decoder(4, 5, "lstm", 1, 4, 4, 4, None, sampling_probability=0.5)
