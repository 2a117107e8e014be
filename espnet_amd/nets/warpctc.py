"""The warp-ctc operator slot with its own calling convention, on the espnet_amd CTC kernels.

reference: `warpctc_pytorch.CTCLoss(size_average, reduce)` as the reference binds it (espnet/nets/pytorch_backend/
ctc.py:40-43,62-63,78-88; espnet2/asr/ctc.py:36-38,53-66):

    loss = ctc_loss(acts, labels, act_lens, label_lens)
        acts       (T, B, V) float32 raw activations on the device (log-softmax is taken inside)
        labels     (sum L) int32 on the CPU: the utterances' labels concatenated, no padding
        act_lens   (B) int32 on the CPU
        label_lens (B) int32 on the CPU
        -> (1,) tensor; sum_b -log p(y_b | x_b), divided by B when size_average; gradients w.r.t. acts only; blank = 0

A trainer that keeps the reference's `CTC` module gets the HIP loss by swapping the import:
`import espnet_amd.nets.warpctc as warp_ctc`.  Our own `CTC` module does not go through here: it hands the padded
int64 labels to the kernel directly and never moves them to the host."""
import numpy as np
import torch

from .. import functional as F_


class CTCLoss(torch.nn.Module):
    def __init__(self, size_average=False, reduce=True, blank=0):
        super().__init__()
        if not reduce:
            raise NotImplementedError("reduce=False (per-utterance losses) is not on the path")
        self.size_average, self.blank = size_average, blank

    def forward(self, acts, labels, act_lens, label_lens):
        if acts.dim() != 3 or acts.dtype != torch.float32 or not acts.is_cuda:
            raise ValueError("acts must be a (T, B, V) float32 device tensor")
        T, B, V = acts.shape
        lab = np.asarray(labels.cpu(), dtype=np.int64).reshape(-1)
        ll = np.asarray(label_lens.cpu(), dtype=np.int64).reshape(-1)
        al = np.asarray(act_lens.cpu(), dtype=np.int32).reshape(-1)
        if ll.size != B or al.size != B or int(ll.sum()) != lab.size:
            raise ValueError("labels / lengths do not describe a batch of %d utterances" % B)
        ys = np.full((B, max(1, int(ll.max()) if B else 1)), -1, dtype=np.int64)
        off = 0
        for b, n in enumerate(ll.tolist()):
            ys[b, :n] = lab[off:off + n]
            off += n
        ys_pad = torch.from_numpy(ys).to(acts.device)
        hl = torch.from_numpy(al).to(acts.device)
        # (T, B, V) is handed over as it is: eamd_ctc_loss takes (stride_t, stride_b) - no 159 MB transposed copy at config 2
        loss = F_.CTCLossFn.apply(acts, ys_pad, hl, self.blank, -1, True)     # sum_b nll_b / B
        return (loss if self.size_average else loss * float(B)).reshape(1)
