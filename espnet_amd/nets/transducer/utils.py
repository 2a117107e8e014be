"""reference: espnet/nets/pytorch_backend/transducer/utils.py:9-53"""
import torch

from ..modules import pad_list


def prepare_loss_inputs(ys_pad, hlens, blank_id=0, ignore_id=-1, device=None):
    """-> ys_in_pad (B,Lmax+1) int64 blank-prefixed, target (B,Lmax) int32, pred_len (B) int32, target_len (B) int32.
    Integer / host work exactly as in the reference (per-utterance Python loops, utils.py:28-51);
    hlens: lengths (list / 1-D tensor) or a (B,1,Tmax) mask."""
    device = ys_pad.device if device is None else torch.device(device)
    ys_cpu = ys_pad.cpu()
    ys = [y[y != ignore_id] for y in ys_cpu]
    blank = ys[0].new([blank_id])
    ys_in = [torch.cat([blank, y], dim=0) for y in ys]
    ys_in_pad = pad_list(ys_in, blank_id)
    target = pad_list(ys, blank_id).type(torch.int32)
    target_len = torch.IntTensor([y.size(0) for y in ys])
    if torch.is_tensor(hlens):
        if hlens.dim() > 1:
            hlens = [int(v) for v in hlens.reshape(hlens.shape[0], -1).ne(0).sum(1).tolist()]
        else:
            hlens = [int(v) for v in hlens.tolist()]
    else:
        hlens = [int(v) for v in hlens]
    pred_len = torch.IntTensor(hlens)
    if device.type != "cuda":
        return ys_in_pad, target, pred_len, target_len
    from ... import ops
    return (ops.h2d_cached("rnnt_ys_in", ys_in_pad.numpy(), device), ops.h2d_cached("rnnt_target", target.numpy(), device),
            ops.h2d_cached("rnnt_tlen", pred_len.numpy(), device), ops.h2d_cached("rnnt_ulen", target_len.numpy(), device))
