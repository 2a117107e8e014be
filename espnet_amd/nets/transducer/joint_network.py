"""reference: espnet/nets/pytorch_backend/transducer/joint_network.py:8-48"""
import torch

from ... import functional as F_
from ... import ops
from ... import rnn_functional as R_

_ACTS = {"tanh": ops.ACT_TANH, "relu": ops.ACT_RELU, "swish": ops.ACT_SWISH}


class JointNetwork(torch.nn.Module):
    """z = lin_out(act(lin_enc(h_enc) + lin_dec(h_dec)))  ->  (B, T, U, vocab_size)"""

    def __init__(self, vocab_size, encoder_output_size, hidden_size, joint_space_size, joint_activation_type):
        super().__init__()
        self.lin_enc = torch.nn.Linear(encoder_output_size, joint_space_size)
        self.lin_dec = torch.nn.Linear(hidden_size, joint_space_size, bias=False)
        self.lin_out = torch.nn.Linear(joint_space_size, vocab_size)
        if joint_activation_type not in _ACTS:
            raise NotImplementedError("joint activation %r has no HIP kernel" % joint_activation_type)
        self.act_id = _ACTS[joint_activation_type]

    def forward(self, h_enc, h_dec):
        """h_enc (B,T,1,D_enc) or (B,T,D_enc); h_dec (B,1,U,D_dec) or (B,U,D_dec)"""
        if h_enc.dim() == 4:
            h_enc = h_enc.squeeze(2)
        if h_dec.dim() == 4:
            h_dec = h_dec.squeeze(1)
        e = F_.LinearFn.apply(h_enc, self.lin_enc.weight, self.lin_enc.bias)
        d = F_.LinearFn.apply(h_dec, self.lin_dec.weight, None)
        h = R_.JointFn.apply(e, d, self.act_id)
        return F_.LinearFn.apply(h, self.lin_out.weight, self.lin_out.bias)
