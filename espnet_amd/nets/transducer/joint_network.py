"""reference: espnet/nets/pytorch_backend/transducer/joint_network.py:8-48"""
import os
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

    def project_enc(self, h):
        """lin_enc over all frames of one utterance, h (T, D_enc) -> (T, J)   (decoding: hoisted out of the search loop)"""
        return F_.LinearFn.apply(h.contiguous(), self.lin_enc.weight, self.lin_enc.bias)

    def joint_step(self, enc_proj_t, y):
        """joint output for one frame and one prediction-network output: enc_proj_t (J,), y (D_dec,) -> (V,)
        == lin_out(act(lin_enc(h_t) + lin_dec(y)))  (joint_network.py:45-46 on 1-D inputs)"""
        d = F_.LinearFn.apply(y.reshape(1, -1).contiguous(), self.lin_dec.weight, None)
        z = R_.JointFn.apply(enc_proj_t.reshape(1, 1, -1).contiguous(), d.reshape(1, 1, -1), self.act_id)
        return F_.LinearFn.apply(z.reshape(1, -1), self.lin_out.weight, self.lin_out.bias).reshape(-1)

    def joint_rows(self, enc_proj_rows, y_rows):
        """decoding on rows: enc_proj_rows (n, J) projected encoder frames, y_rows (n, D_dec) or (1, D_dec) prediction
        outputs (one output against every frame) -> logits (n, V): one lin_dec GEMM, one fused add + activation, one
        lin_out GEMM for all n (frame, hypothesis) pairs"""
        n = enc_proj_rows.shape[0]
        d = F_.LinearFn.apply(y_rows.contiguous(), self.lin_dec.weight, None)
        if d.shape[0] == 1:
            z = R_.JointFn.apply(enc_proj_rows.contiguous().view(1, n, -1), d.view(1, 1, -1), self.act_id)
        else:
            z = R_.JointFn.apply(enc_proj_rows.contiguous().view(n, 1, -1), d.view(n, 1, -1), self.act_id)
        return F_.LinearFn.apply(z.reshape(n, -1), self.lin_out.weight, self.lin_out.bias)

    def loss(self, h_enc, h_dec, target, pred_len, target_len, pred_len_host, blank, chunk_rows=None):
        """transducer loss from encoder states (B,T,D_enc) and prediction-network states (B,U,D_dec) without the
        (B,T,U,V) logits (rnn_functional.JointRNNTLossFn): mean over the batch of -log P(y | x)"""
        if chunk_rows is None:
            chunk_rows = int(os.environ.get("EAMD_RNNT_CHUNK_ROWS", str(1 << 16)))
        e = F_.LinearFn.apply(h_enc, self.lin_enc.weight, self.lin_enc.bias)
        d = F_.LinearFn.apply(h_dec, self.lin_dec.weight, None)
        return R_.JointRNNTLossFn.apply(e, d, self.lin_out.weight, self.lin_out.bias, target.contiguous(), pred_len.contiguous(),
                                        target_len.contiguous(), blank, self.act_id, [int(v) for v in pred_len_host], chunk_rows)

    def forward(self, h_enc, h_dec):
        """h_enc (B,T,1,D_enc) or (B,T,D_enc); h_dec (B,1,U,D_dec) or (B,U,D_dec)"""
        if h_enc.dim() == 1 and h_dec.dim() == 1:      # decoding: one frame, one prediction-network output
            return self.joint_step(self.project_enc(h_enc.reshape(1, -1))[0], h_dec)
        if h_enc.dim() == 4:
            h_enc = h_enc.squeeze(2)
        if h_dec.dim() == 4:
            h_dec = h_dec.squeeze(1)
        e = F_.LinearFn.apply(h_enc, self.lin_enc.weight, self.lin_enc.bias)
        d = F_.LinearFn.apply(h_dec, self.lin_dec.weight, None)
        h = R_.JointFn.apply(e, d, self.act_id)
        return F_.LinearFn.apply(h, self.lin_out.weight, self.lin_out.bias)
