"""autograd Functions of the RNN paths (VGG-BLSTMP encoder, location-aware attention decoder,
RNN-Transducer predictor / joint network / loss): explicit HIP kernel sequences over the C ABI, same
conventions as espnet_amd.functional (weight gradients go to the flat gradient arena when present).

reference: rnn/encoders.py, rnn/attentions.py:250-380, rnn/decoders.py:142-311,
transducer/rnn_decoder.py, transducer/joint_network.py, transducer/loss.py.
"""
import torch

from . import ops
from .functional import GradSink
from .ops import ACT_RELU, EPI_MUL_RELU_MASK, EPI_RELU


class ActFn(torch.autograd.Function):
    """elementwise activation (tanh between RNNP layers, encoders.py:98)"""

    @staticmethod
    def forward(ctx, x, act):
        x = x.contiguous()
        ctx.save_for_backward(x)
        ctx.act = act
        return ops.act_fwd_any(x, act)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.act_bwd_any(dy.contiguous(), x, ctx.act), None


class PlainEmbedFn(torch.autograd.Function):
    """torch.nn.Embedding lookup; pad_idx >= 0 = Embedding(padding_idx): no gradient for that row
    (rnn/decoders.py:88, transducer/rnn_decoder.py:44)."""

    @staticmethod
    def forward(ctx, tokens, table, pad_idx):
        tok = tokens.contiguous()
        out = ops.embed_pe(tok.view(-1), table, None, 1, 1.0)
        ctx.save_for_backward(tok)
        ctx.pr = GradSink.use((table,))
        ctx.pad = pad_idx
        return out.view(*tokens.shape, table.shape[1])

    @staticmethod
    def backward(ctx, dout):
        (tok,) = ctx.saved_tensors
        sink = GradSink(ctx.pr)
        ops.embed_bwd(tok.view(-1), dout.reshape(-1, dout.shape[-1]).contiguous(), sink.buf(0), 1.0, ctx.pad)
        return (None,) + sink.results() + (None,)


# =================================================================================================
# LSTM over a whole sequence whose input projections are known up front (encoder layers, transducer
# predictor): gx[T,B,4H] = x W_ih^T + b_ih for all frames is ONE GEMM done by the caller; this Function
# runs the recurrence (one small GEMM + one pointwise kernel per frame) and, in backward, gathers the
# recurrent weight gradient of all frames into a single GEMM.
# live[T,B] uint8 reproduces pack_padded_sequence: finished sequences keep their state, emit zeros.
# =================================================================================================
def _rec_gemm(a, w, out, M, N, K, lda, ldb, ldc, *, transB=0, bias=None, R=None, ldr=0):
    """recurrent-step product with a handful of rows (M = batch): one row of 64-wide tiles leaves most CUs idle and
    each tile walks the whole reduction alone, so the reduction is split across workgroups (f32 atomics into the
    zeroed result; bias / R are added by the leading split)"""
    tiles = ((M + 63) // 64) * ((N + 63) // 64)
    sk = 1
    while tiles * sk < 512 and K // (sk * 2) >= 128:
        sk *= 2
    if out is None:        # a fresh [M, N] result: zeros come from the step's zero arena (no fill launch)
        out = ops.zeros(M, N, device=a.device) if sk > 1 else torch.empty(M, N, device=a.device, dtype=torch.float32)
    elif sk > 1:
        out.zero_()
    ops.gemm(a, w, out, M, N, K, lda, ldb, ldc, transB=transB, bias=bias, R=R, ldr=ldr, splitk=sk)
    return out


class LSTMSeqFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gx, w_hh, b_hh, live, reverse):
        T, B, H4 = gx.shape
        H = H4 // 4
        dev = gx.device
        gx = gx.contiguous()
        h_out = torch.empty(T, B, H, device=dev, dtype=torch.float32)
        c_out = torch.empty(T, B, H, device=dev, dtype=torch.float32)
        acts = torch.empty(T, B, H4, device=dev, dtype=torch.float32)
        y = torch.empty(T, B, H, device=dev, dtype=torch.float32)
        zero = torch.zeros(B, H, device=dev, dtype=torch.float32)
        gates = torch.empty(B, H4, device=dev, dtype=torch.float32)
        hp, cp = zero, zero
        fused = ops.lstm_step_ok(B, H)       # recurrent product + cell update of a step as ONE launch
        for t in (range(T - 1, -1, -1) if reverse else range(T)):
            if fused:
                ops.lstm_step_fwd(gx[t], w_hh, b_hh, hp, cp, None if live is None else live[t], h_out[t], c_out[t], y[t], acts[t])
            else:
                _rec_gemm(hp, w_hh, gates, B, H4, H, H, H, H4, bias=b_hh, R=gx[t], ldr=H4)
                ops.lstm_cell_fwd(gates, cp, hp, None if live is None else live[t], h_out[t], c_out[t], y[t], acts[t])
            hp, cp = h_out[t], c_out[t]
        ctx.save_for_backward(h_out, c_out, acts, zero, live if live is not None else zero)
        ctx.pr = GradSink.use((w_hh, b_hh))
        ctx.cfg = (reverse, live is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        h_out, c_out, acts, zero, live = ctx.saved_tensors
        w_hh, b_hh = ctx.pr
        reverse, has_live = ctx.cfg
        T, B, H4 = acts.shape
        H = H4 // 4
        dev = acts.device
        dy = dy.contiguous()
        sink = GradSink(ctx.pr)
        dgates = torch.empty(T, B, H4, device=dev, dtype=torch.float32)
        dh, dc = None, None
        if ops.lstm_step_ok(B, H):
            # one launch per step: dh = dgates_next W_hh (+ the masked pass-through) and the cell backward; W_hh^T is made
            # once per sequence so that the product's reduction runs along contiguous memory
            w_t = w_hh.detach().t().contiguous()
            dg_next = dpass = None
            for t in (range(T) if reverse else range(T - 1, -1, -1)):
                pt = t + 1 if reverse else t - 1
                cprev = c_out[pt] if 0 <= pt < T else zero
                dc_new = torch.empty(B, H, device=dev, dtype=torch.float32)
                dpass_new = torch.empty(B, H, device=dev, dtype=torch.float32)
                ops.lstm_step_bwd(dy[t], dg_next, w_t, dpass, dc, acts[t], cprev, c_out[t], live[t] if has_live else None,
                                  dgates[t], dc_new, dpass_new)
                dg_next, dpass, dc = dgates[t], dpass_new, dc_new
            if T > 1:
                dg, hp = (dgates[:-1], h_out[1:]) if reverse else (dgates[1:], h_out[:-1])
                ops.linear_bwd_w(dg.reshape(-1, H4), hp.reshape(-1, H), sink.buf(0))
            ops.colsum(dgates.view(-1, H4), sink.buf(1))
            return (dgates,) + sink.results() + (None, None)
        for t in (range(T) if reverse else range(T - 1, -1, -1)):
            pt = t + 1 if reverse else t - 1            # frame whose state fed this step
            has_prev = 0 <= pt < T
            cprev = c_out[pt] if has_prev else zero
            dc_new = torch.empty(B, H, device=dev, dtype=torch.float32)
            dh_pass = torch.empty(B, H, device=dev, dtype=torch.float32) if has_live else None
            ops.lstm_cell_bwd(dy[t], dh, dc, acts[t], cprev, c_out[t], live[t] if has_live else None, dgates[t],
                              dc_new, dh_pass)
            if has_prev:
                dh = torch.empty(B, H, device=dev, dtype=torch.float32)
                _rec_gemm(dgates[t], w_hh, dh, B, H, H4, H4, H, H, transB=1, R=dh_pass, ldr=H)
            dc = dc_new
        if T > 1:
            dg, hp = (dgates[:-1], h_out[1:]) if reverse else (dgates[1:], h_out[:-1])
            ops.linear_bwd_w(dg.reshape(-1, H4), hp.reshape(-1, H), sink.buf(0))
        ops.colsum(dgates.view(-1, H4), sink.buf(1))
        return (dgates,) + sink.results() + (None, None)


class LSTMSeqGroupFn(torch.autograd.Function):
    """One or two independent LSTM recurrences over the same (T, B) - the directions of a BLSTM layer - as ONE
    persistent launch forward and ONE backward (csrc/lstm_seq.hip).  apply(live, n, gx_0, w_hh_0, b_hh_0, rev_0, ...)
    -> (y_0, ...).  reference: rnn/encoders.py:36-39,110-117 (torch.nn.LSTM(bidirectional=True) on packed sequences)."""

    @staticmethod
    def forward(ctx, live, n, *flat):
        jobs = [flat[4 * i: 4 * i + 4] for i in range(n)]
        T, B, H4 = jobs[0][0].shape
        H = H4 // 4
        dev = jobs[0][0].device
        saved, outs, launch = [], [], []
        for gx, w_hh, b_hh, rev in jobs:
            gx = gx.contiguous()
            h_out = torch.empty(T, B, H, device=dev, dtype=torch.float32)
            c_out = torch.empty(T, B, H, device=dev, dtype=torch.float32)
            acts = torch.empty(T, B, H4, device=dev, dtype=torch.float32)
            y = torch.empty(T, B, H, device=dev, dtype=torch.float32)
            launch.append((gx, w_hh, b_hh, live, h_out, c_out, y, acts, rev))
            saved += [h_out, c_out, acts]
            outs.append(y)
        ops.lstm_seq_fwd(launch, T, B, H)
        ctx.save_for_backward(*saved, *([live] if live is not None else []))
        ctx.prs = [GradSink.use((w_hh, b_hh)) for _gx, w_hh, b_hh, _r in jobs]
        ctx.cfg = (n, [bool(j[3]) for j in jobs], live is not None)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dys):
        n, revs, has_live = ctx.cfg
        saved = ctx.saved_tensors
        live = saved[3 * n] if has_live else None
        T, B, H4 = saved[2].shape
        H = H4 // 4
        dev = saved[2].device
        launch, dgs = [], []
        for i in range(n):
            h_out, c_out, acts = saved[3 * i: 3 * i + 3]
            w_hh, _b = ctx.prs[i]
            dy = dys[i].contiguous() if dys[i] is not None else torch.zeros(T, B, H, device=dev, dtype=torch.float32)
            dgates = torch.empty(T, B, H4, device=dev, dtype=torch.float32)
            launch.append((dy, w_hh.detach().t().contiguous(), acts, c_out, live, dgates, revs[i]))
            dgs.append(dgates)
        ops.lstm_seq_bwd(launch, T, B, H)
        res = [None, None]
        for i in range(n):
            h_out = saved[3 * i]
            sink = GradSink(ctx.prs[i])
            if T > 1:
                dg, hp = (dgs[i][:-1], h_out[1:]) if revs[i] else (dgs[i][1:], h_out[:-1])
                ops.linear_bwd_w(dg.reshape(-1, H4), hp.reshape(-1, H), sink.buf(0))
            ops.colsum(dgs[i].view(-1, H4), sink.buf(1))
            res += [dgs[i]] + list(sink.results()) + [None]
        return tuple(res)


class LSTMCellFn(torch.autograd.Function):
    """one LSTMCell step whose input depends on the previous step (attention decoder, decoders.py:120-134):
    gates = gx (= x W_ih^T + b_ih, a LinearFn product) + h W_hh^T + b_hh."""

    @staticmethod
    def forward(ctx, gx, h_prev, c_prev, w_hh, b_hh):
        B, H4 = gx.shape
        H = H4 // 4
        dev = gx.device
        gx, h_prev, c_prev = gx.contiguous(), h_prev.contiguous(), c_prev.contiguous()
        gates = torch.empty(B, H4, device=dev, dtype=torch.float32)
        h = torch.empty(B, H, device=dev, dtype=torch.float32)
        c = torch.empty(B, H, device=dev, dtype=torch.float32)
        acts = gates                                       # activated gates overwrite the pre-activations
        if ops.lstm_step_ok(B, H):
            ops.lstm_step_fwd(gx, w_hh, b_hh, h_prev, c_prev, None, h, c, None, acts)
        else:
            _rec_gemm(h_prev, w_hh, gates, B, H4, H, H, H, H4, bias=b_hh, R=gx, ldr=H4)
            ops.lstm_cell_fwd(gates, c_prev, None, None, h, c, None, acts)
        ctx.save_for_backward(acts, h_prev, c_prev, c)
        ctx.pr = GradSink.use((w_hh, b_hh) if b_hh is not None else (w_hh,))      # LSTMCell(bias=False): AttLocRec's att_lstm
        return h, c

    @staticmethod
    def backward(ctx, dh, dc):
        acts, h_prev, c_prev, c = ctx.saved_tensors
        w_hh = ctx.pr[0]
        B, H4 = acts.shape
        H = H4 // 4
        dev = acts.device
        sink = GradSink(ctx.pr)
        dgates = torch.empty(B, H4, device=dev, dtype=torch.float32)
        dc_prev = torch.empty(B, H, device=dev, dtype=torch.float32)
        ops.lstm_cell_bwd(None, dh.contiguous() if dh is not None else None, dc.contiguous() if dc is not None else None,
                          acts, c_prev, c, None, dgates, dc_prev, None)
        dh_prev = _rec_gemm(dgates, w_hh, None, B, H, H4, H4, H, H, transB=1)
        has_b = len(ctx.pr) > 1
        ops.linear_bwd_w(dgates, h_prev, sink.buf(0), db=sink.buf(1) if has_b else None)
        return (dgates, dh_prev, dc_prev) + sink.results() + (() if has_b else (None,))


# =================================================================================================
# GRU: same structure as the LSTM Functions; the recurrent product keeps its own [B,3H] buffer because the
# candidate gate needs r * (h W_hn^T + b_hn), not a plain sum with the input projection.
# =================================================================================================
class GRUSeqFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gx, w_hh, b_hh, live, reverse):
        T, B, H3 = gx.shape
        H = H3 // 3
        dev = gx.device
        gx = gx.contiguous()
        h_out = torch.empty(T, B, H, device=dev, dtype=torch.float32)
        acts = torch.empty(T, B, 4 * H, device=dev, dtype=torch.float32)
        y = torch.empty(T, B, H, device=dev, dtype=torch.float32)
        zero = torch.zeros(B, H, device=dev, dtype=torch.float32)
        gh = torch.empty(B, H3, device=dev, dtype=torch.float32)
        hp = zero
        for t in (range(T - 1, -1, -1) if reverse else range(T)):
            _rec_gemm(hp, w_hh, gh, B, H3, H, H, H, H3, bias=b_hh)
            ops.gru_cell_fwd(gx[t], gh, hp, None if live is None else live[t], h_out[t], y[t], acts[t])
            hp = h_out[t]
        ctx.save_for_backward(h_out, acts, zero, live if live is not None else zero)
        ctx.pr = GradSink.use((w_hh, b_hh))
        ctx.cfg = (reverse, live is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        h_out, acts, zero, live = ctx.saved_tensors
        w_hh, b_hh = ctx.pr
        reverse, has_live = ctx.cfg
        T, B, H4 = acts.shape
        H = H4 // 4
        dev = acts.device
        dy = dy.contiguous()
        sink = GradSink(ctx.pr)
        dgx = torch.empty(T, B, 3 * H, device=dev, dtype=torch.float32)
        dgh = torch.empty(T, B, 3 * H, device=dev, dtype=torch.float32)
        dh = None
        for t in (range(T) if reverse else range(T - 1, -1, -1)):
            pt = t + 1 if reverse else t - 1
            has_prev = 0 <= pt < T
            hprev = h_out[pt] if has_prev else zero
            direct = torch.empty(B, H, device=dev, dtype=torch.float32)
            ops.gru_cell_bwd(dy[t], dh, acts[t], hprev, live[t] if has_live else None, dgx[t], dgh[t], direct)
            if has_prev:
                dh = torch.empty(B, H, device=dev, dtype=torch.float32)
                _rec_gemm(dgh[t], w_hh, dh, B, H, 3 * H, 3 * H, H, H, transB=1, R=direct, ldr=H)
        if T > 1:
            dg, hp = (dgh[:-1], h_out[1:]) if reverse else (dgh[1:], h_out[:-1])
            ops.linear_bwd_w(dg.reshape(-1, 3 * H), hp.reshape(-1, H), sink.buf(0))
        ops.colsum(dgh.view(-1, 3 * H), sink.buf(1))
        return (dgx,) + sink.results() + (None, None)


class GRUCellFn(torch.autograd.Function):
    """one GRUCell step (decoders whose input depends on the previous step); gx = x W_ih^T + b_ih"""

    @staticmethod
    def forward(ctx, gx, h_prev, w_hh, b_hh):
        B, H3 = gx.shape
        H = H3 // 3
        dev = gx.device
        gx, h_prev = gx.contiguous(), h_prev.contiguous()
        gh = torch.empty(B, H3, device=dev, dtype=torch.float32)
        _rec_gemm(h_prev, w_hh, gh, B, H3, H, H, H, H3, bias=b_hh)
        h = torch.empty(B, H, device=dev, dtype=torch.float32)
        acts = torch.empty(B, 4 * H, device=dev, dtype=torch.float32)
        ops.gru_cell_fwd(gx, gh, h_prev, None, h, None, acts)
        ctx.save_for_backward(acts, h_prev)
        ctx.pr = GradSink.use((w_hh, b_hh))
        return h

    @staticmethod
    def backward(ctx, dh):
        acts, h_prev = ctx.saved_tensors
        w_hh, b_hh = ctx.pr
        B, H4 = acts.shape
        H = H4 // 4
        dev = acts.device
        sink = GradSink(ctx.pr)
        dgx = torch.empty(B, 3 * H, device=dev, dtype=torch.float32)
        dgh = torch.empty(B, 3 * H, device=dev, dtype=torch.float32)
        direct = torch.empty(B, H, device=dev, dtype=torch.float32)
        ops.gru_cell_bwd(None, dh.contiguous(), acts, h_prev, None, dgx, dgh, direct)
        dh_prev = torch.empty(B, H, device=dev, dtype=torch.float32)
        _rec_gemm(dgh, w_hh, dh_prev, B, H, 3 * H, 3 * H, H, H, transB=1, R=direct, ldr=H)
        ops.linear_bwd_w(dgh, h_prev, sink.buf(0), db=sink.buf(1))
        return (dgx, dh_prev) + sink.results()


# =================================================================================================
# VGG2L front-end: 2 x (conv3x3-ReLU, conv3x3-ReLU, maxpool 2x2 ceil) on NHWC activations.
# reference: rnn/encoders.py:178-237.  conv1_1 (C_in = 1) is a direct kernel, the other three are
# implicit GEMMs over the gather descriptor (zero padding = out-of-range taps), ReLU fused in the
# epilogue; the ReLU masks of the backward are fused into the input-gradient GEMMs.
# =================================================================================================
_TAPS = [(kh, kw) for kh in range(3) for kw in range(3)]
# tap order of the input-gradient weight image built by eamd_conv2_weight_prep (misc.hip: kTapOrder)
_TAPS_DX = [(0, 0), (0, 2), (2, 0), (2, 2), (0, 1), (2, 1), (1, 0), (1, 2), (1, 1)]


def _conv3x3_fwd(x, w, b, B, H, W):
    """x NHWC [B,H,W,Ci] -> relu(conv3x3 pad 1) [B,H,W,Co]"""
    Co, Ci = w.shape[0], w.shape[1]
    wf, wd = ops.conv2_weight_prep(w, x.dtype)
    g = ops.make_gather(Ci, [(kh - 1, kw - 1) for kh, kw in _TAPS], H, W, H, W, 1, 1)
    M = B * H * W
    y = torch.empty(M, Co, device=x.device, dtype=x.dtype)
    ops.gemm(x, wf, y, M, Co, 9 * Ci, 9 * Ci, Co, Co, transB=1, bias=b, epilogue=EPI_RELU, gather=g)
    return y.view(B, H, W, Co), wd


def _conv3x3_bwd(dy, x, wd, dw_buf, db_buf, B, H, W, Co, Ci, relu_aux):
    """dy [B*H*W, Co] (already masked by this conv's ReLU); accumulates dW, db; returns dX
    (masked by relu_aux > 0 when given = ReLU of the producing layer)"""
    M = B * H * W
    dev = dy.device
    ops.colsum(dy, db_buf)
    dwf = torch.zeros(9 * Ci, Co, device=dev, dtype=torch.float32)
    g = ops.make_gather(Ci, [(kh - 1, kw - 1) for kh, kw in _TAPS], H, W, H, W, 1, 1)
    ntile = (9 * Ci // 64) * ((Co + 63) // 64)
    sk = max(1, min(64, (1024 + ntile - 1) // ntile, max(1, M // 256)))
    ops.gemm(x, dy, dwf, 9 * Ci, Co, M, 9 * Ci, Co, Co, transA=1, transB=1, gather=g, splitk=sk, tile=64)
    ops.conv2_weight_grad(dwf, dw_buf, Co, Ci)
    gt = ops.make_gather(Co, [(1 - kh, 1 - kw) for kh, kw in _TAPS_DX], H, W, H, W, 1, 1)
    dx = torch.empty(M, Ci, device=dev, dtype=dy.dtype)
    if relu_aux is not None:
        ops.gemm(dy, wd, dx, M, Ci, 9 * Co, 9 * Co, Ci, Ci, transB=1, gather=gt, epilogue=EPI_MUL_RELU_MASK,
                 aux=relu_aux, ldaux=Ci)
    else:
        ops.gemm(dy, wd, dx, M, Ci, 9 * Co, 9 * Co, Ci, Ci, transB=1, gather=gt)
    return dx


class VGG2LFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w11, b11, w12, b12, w21, b21, w22, b22):
        B, T, F = x.shape
        x = x.contiguous()
        C1, C2 = w11.shape[0], w21.shape[0]
        y1 = ops.conv3x3_c1_fwd(x, w11, b11, B, T, F, C1)                     # [B,T,F,64]
        y2, wd12 = _conv3x3_fwd(y1, w12, b12, B, T, F)
        p1, i1 = ops.maxpool2x2_fwd(y2)
        T2, F2 = p1.shape[1], p1.shape[2]
        y3, wd21 = _conv3x3_fwd(p1, w21, b21, B, T2, F2)
        y4, wd22 = _conv3x3_fwd(y3, w22, b22, B, T2, F2)
        p2, i2 = ops.maxpool2x2_fwd(y4)
        T4, F4 = p2.shape[1], p2.shape[2]
        out = torch.empty(B, T4, C2, F4, device=x.device, dtype=torch.float32)   # (c, f) feature order
        ops.permute4(p2, out, (B * T4, F4, C2, 1), (C2 * F4, 1, F4, 0))
        ctx.save_for_backward(x, y1, y2, i1, p1, y3, y4, i2, wd12, wd21, wd22)
        ctx.pr = GradSink.use((w11, b11, w12, b12, w21, b21, w22, b22))
        ctx.cfg = (B, T, F, T2, F2, T4, F4, C1, C2)
        return out.view(B, T4, C2 * F4)

    @staticmethod
    def backward(ctx, dout):
        x, y1, y2, i1, p1, y3, y4, i2, wd12, wd21, wd22 = ctx.saved_tensors
        B, T, F, T2, F2, T4, F4, C1, C2 = ctx.cfg
        sink = GradSink(ctx.pr)
        dev = x.device
        dp2 = torch.empty(B, T4, F4, C2, device=dev, dtype=torch.float32)
        ops.permute4(dout.contiguous(), dp2, (B * T4, C2, F4, 1), (F4 * C2, 1, C2, 0))
        dy4 = ops.maxpool2x2_bwd(dp2, i2, (B, T2, F2, C2))
        dy4 = ops.act_bwd_any(dy4, y4, ACT_RELU)                               # y4 > 0 <=> pre-activation > 0
        dy3 = _conv3x3_bwd(dy4.view(-1, C2), y3, wd22, sink.buf(6), sink.buf(7), B, T2, F2, C2, C2, y3.view(-1, C2))
        dp1 = _conv3x3_bwd(dy3, p1, wd21, sink.buf(4), sink.buf(5), B, T2, F2, C2, C1, None)
        dy2 = ops.maxpool2x2_bwd(dp1.view(B, T2, F2, C1), i1, (B, T, F, C1))
        dy2 = ops.act_bwd_any(dy2, y2, ACT_RELU)
        dy1 = _conv3x3_bwd(dy2.view(-1, C1), y1, wd12, sink.buf(2), sink.buf(3), B, T, F, C1, C1, y1.view(-1, C1))
        ops.conv3x3_c1_bwd_w(dy1, x, sink.buf(0), sink.buf(1), B, T, F, C1)
        return (None,) + sink.results()


# =================================================================================================
# Location-aware attention, one decoder step (AttLoc.forward, attentions.py:300-380)
# inputs: enc_h [B,T,E], pre_enc = mlp_enc(enc_h) [B,T,A], dec_proj = mlp_dec(z) [B,A], att_prev [B,T]
# params: loc_conv.weight [C,1,1,K], mlp_att.weight [A,C], gvec.weight [1,A], gvec.bias [1]
# =================================================================================================
class AttLocStepFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, enc_h, pre_enc, dec_proj, att_prev, lens, scaling, conv_w, w_att, gvec_w, gvec_b, acc=None,
                first=False):
        # acc / first (AttLoc): enc_h and pre_enc are the same tensors at every decoder step, so their gradients are one
        # running sum kept in `acc` by the step kernels; only the FIRST step - whose backward runs last: every later
        # step hangs on it through the attention weights and the decoder state - hands that sum to autograd, the others
        # return None (101 fresh 33 MB tensors each and autograd's adds otherwise: 4 ms per step at config 4)
        ctx.acc, ctx.first = acc, first
        enc_h, pre_enc, dec_proj = (t.contiguous() for t in (enc_h, pre_enc, dec_proj))
        att_prev = att_prev.contiguous() if conv_w is not None else None       # conv_w None: additive attention
        c, w, th, conv = ops.attloc_fwd(att_prev, conv_w, w_att, pre_enc, dec_proj, gvec_w, gvec_b, lens, enc_h,
                                        scaling)
        ctx.save_for_backward(enc_h, att_prev, w, th, conv)
        ctx.pr = GradSink.use((conv_w, w_att, gvec_w, gvec_b) if conv_w is not None else (gvec_w, gvec_b))
        ctx.has_conv = conv_w is not None
        ctx.scaling = scaling
        return c, w

    @staticmethod
    def backward(ctx, dc, dw):
        enc_h, att_prev, w, th, conv = ctx.saved_tensors
        sink = GradSink(ctx.pr)
        B, T, A = th.shape
        if not ctx.has_conv:
            gvec_w, gvec_b = ctx.pr
            d_enc_h, df, d_dec = ops.attloc_bwd_energy(dc.contiguous(), dw.contiguous() if dw is not None else None,
                                                       w, enc_h, th, gvec_w, ctx.scaling, sink.buf(0), sink.buf(1))
            return (d_enc_h, df, d_dec, None, None, None, None, None) + sink.results() + (None, None)
        conv_w, w_att, gvec_w, gvec_b = ctx.pr
        Cc = conv.shape[2]
        dcc, dwc = dc.contiguous(), dw.contiguous() if dw is not None else None
        # one pass over th forms df, dconv = df @ W_att and dW_att += df^T conv (two N = C GEMMs otherwise)
        acc = ctx.acc
        r = ops.attloc_bwd_energy_conv(dcc, dwc, w, enc_h, th, gvec_w, ctx.scaling, conv, w_att, sink.buf(2), sink.buf(3),
                                       sink.buf(1), acc=acc)
        if r is not None:
            d_enc_h, df, d_dec, dconv = r
            if acc is not None:
                if ctx.first:          # the last backward of the sequence: the sums go to autograd
                    acc.clear()
                else:
                    d_enc_h = df = None
        else:
            d_enc_h, df, d_dec = ops.attloc_bwd_energy(dcc, dwc, w, enc_h, th, gvec_w, ctx.scaling, sink.buf(2), sink.buf(3))
            df2 = df.view(B * T, A)
            dconv = torch.empty(B * T, Cc, device=df.device, dtype=torch.float32)
            ops.gemm(df2, w_att, dconv, B * T, Cc, A, A, Cc, Cc, transB=1)              # dconv = df @ W_att
            ops.linear_bwd_w(df2, conv.view(B * T, Cc), sink.buf(1))                     # dW_att += df^T conv
        d_prev = ops.attloc_bwd_conv(dconv.view(B, T, Cc), conv_w, att_prev, sink.buf(0))
        return (d_enc_h, df, d_dec, d_prev, None, None) + sink.results() + (None, None)


class AttDotStepFn(torch.autograd.Function):
    """dot-product attention step on activated keys / query (AttDot, AttMultiHeadDot heads):
    e = k . q, w = softmax(scaling * e) over the valid frames, ctx = sum_t w * v"""

    @staticmethod
    def forward(ctx, k, q, v, lens, scaling):
        k, q, v = k.contiguous(), q.contiguous(), v.contiguous()
        c, w = ops.att_dot_fwd(k, q, v, lens, scaling)
        ctx.save_for_backward(k, q, v, w)
        ctx.scaling = scaling
        return c, w

    @staticmethod
    def backward(ctx, dc, dw):
        k, q, v, w = ctx.saved_tensors
        d_v, dk, dq = ops.att_dot_bwd(dc.contiguous(), dw.contiguous() if dw is not None else None, w, k, q, v,
                                      ctx.scaling)
        return dk, dq, d_v, None, None


class ConvMaxFn(torch.autograd.Function):
    """max over frames of relu(Conv2d(1, C, (1, K))(att_prev)) (AttLocRec, attentions.py:690-696) -> [B, C]"""

    @staticmethod
    def forward(ctx, att_prev, conv_w):
        att_prev = att_prev.contiguous()
        pooled, idx = ops.attloc_convmax_fwd(att_prev, conv_w)
        ctx.save_for_backward(att_prev, pooled, idx)
        ctx.pr = GradSink.use((conv_w,))
        return pooled

    @staticmethod
    def backward(ctx, dpool):
        att_prev, pooled, idx = ctx.saved_tensors
        sink = GradSink(ctx.pr)
        d_prev = ops.attloc_convmax_bwd(dpool.contiguous(), pooled, idx, att_prev, ctx.pr[0], sink.buf(0))
        return (d_prev,) + sink.results()


class AddFn(torch.autograd.Function):
    """a + b (running coverage vector of AttCov / AttCovLoc, attentions.py:433,795)"""

    @staticmethod
    def forward(ctx, a, b):
        return ops.axpby(a.contiguous(), b.contiguous(), 1.0, 1.0)

    @staticmethod
    def backward(ctx, g):
        return g, g


# =================================================================================================
# Transducer joint network pointwise part and loss
# =================================================================================================
class JointFn(torch.autograd.Function):
    """h[b,t,u,:] = act(enc_proj[b,t,:] + dec_proj[b,u,:])   (joint_network.py:45)"""

    @staticmethod
    def forward(ctx, enc_proj, dec_proj, act):
        enc_proj, dec_proj = enc_proj.contiguous(), dec_proj.contiguous()
        ctx.save_for_backward(enc_proj, dec_proj)
        ctx.act = act
        return ops.joint_fwd(enc_proj, dec_proj, act)

    @staticmethod
    def backward(ctx, dh):
        enc_proj, dec_proj = ctx.saved_tensors
        d_enc, d_dec = ops.joint_bwd(dh.contiguous(), enc_proj, dec_proj, ctx.act)
        return d_enc, d_dec, None


class RNNTLossFn(torch.autograd.Function):
    """mean_b -log P(y_b | x_b) on raw joint logits [B,T,U,V] (transducer/loss.py:74-76: warp-transducer
    RNNTLoss(blank), default reduction = mean over the batch).  Forward fills the lattice workspace (lse, the two
    log-probabilities per node, alpha, beta); backward is ONE pass over the logits that writes the gradient with
    the upstream scalar (read on the device) already folded in."""

    @staticmethod
    def forward(ctx, logits, labels, tlens, ulens, blank):
        B = logits.shape[0]
        logits = logits.contiguous()
        nll, ws = ops.rnnt_loss(logits, labels, tlens, ulens, blank, return_ws=True)
        ctx.save_for_backward(logits, labels, tlens, ulens, ws)
        ctx.blank = blank
        ctx.nll = nll
        return ops.reduce_sum(nll, 1.0 / B)

    @staticmethod
    def backward(ctx, g):
        logits, labels, tlens, ulens, ws = ctx.saved_tensors
        B = logits.shape[0]
        grad = ops.rnnt_grad(logits, labels, tlens, ulens, ctx.blank, ws, g.reshape(1).contiguous(), 1.0 / B)
        return grad, None, None, None, None


class JointRNNTLossFn(torch.autograd.Function):
    """mean_b -log P(y_b | x_b) straight from the joint network's two projections, WITHOUT the (B, T, U, V) logits:
        z[b,t,u,:] = lin_out(act(e[b,t,:] + d[b,u,:]))      e = lin_enc(h_enc) (B,T,J), d = lin_dec(h_dec) (B,U,J)
    reference: transducer/joint_network.py:34-48 + transducer/rnn_decoder.py:160-165 + transducer/loss.py:74-76, which
    materialise z (12 GB at B=16, T'=374, U=101, V=5000) and hand it to warp-transducer.

    Here the lattice is streamed in row chunks of one utterance's frames [t0, t1) x all U (at most `chunk_rows` nodes):
      forward : H = act(e + d) for the chunk -> Z = H W_out^T + b (one GEMM) -> per-node lse and the two log-probabilities
                the lattice needs (eamd_rnnt_node_stats) -> Z is dropped; then alpha / beta on the (B,T,U) lattices;
      backward: H and Z of the chunk are recomputed, eamd_rnnt_node_grad turns Z into dZ in place (upstream scalar read
                on the device), dH = dZ W_out, dW_out += dZ^T H, db_out += colsum(dZ), and the joint add's two
                reductions give de rows and the utterance's dd.
    Frames past an utterance's length are never touched.  Live memory: one chunk of logits (chunk_rows x V) + the
    lattice workspace, instead of logits + their gradient + the (B,T,U,J) joint activations and their gradient."""

    @staticmethod
    def forward(ctx, e, d, w_out, b_out, labels, tlens, ulens, blank, act, tlens_host, chunk_rows):
        B, T, J = e.shape
        U = d.shape[1]
        e, d = e.contiguous(), d.contiguous()
        ws = ops.rnnt_workspace(B, T, U, e.device)
        adt = ops.act_dtype()
        nt_max = max(1, int(chunk_rows) // U)
        plan = [(b, t0, min(nt_max, int(tlens_host[b]) - t0)) for b in range(B) for t0 in range(0, int(tlens_host[b]), nt_max)]
        lab_u = torch.cat([labels.view(B, U - 1), labels.new_full((B, 1), -1)], 1) if U > 1 else labels.new_full((B, 1), -1)
        for b, t0, nt in plan:
            H = ops.joint_fwd(e[b:b + 1, t0:t0 + nt], d[b:b + 1], act, out_dtype=adt)             # (1, nt, U, J)
            # the logits GEMM leaves per-tile softmax partials and the two logits a node needs instead of the chunk of
            # logits (378-756 MB written and read back per chunk at config 5); declined shapes store the logits
            col = lab_u[b].repeat(nt).contiguous()           # next label of node (t, u), -1 in the last column
            if not ops.rnnt_node_stats_fused(H.view(nt * U, J), ops.wshadow(w_out), b_out, col, ws, (b * T + t0) * U, B, T, U,
                                             blank):
                Z = ops.linear_fwd(H.view(nt * U, J), ops.wshadow(w_out), b_out)
                ops.rnnt_node_stats(Z, labels, ws, (b * T + t0) * U, B, T, U, blank)
        nll = ops.rnnt_alpha_beta(ws, tlens, ulens, B, T, U)
        ctx.save_for_backward(e, d, labels, tlens, ulens, ws)
        ctx.pr = GradSink.use((w_out, b_out))
        ctx.cfg = (blank, act, plan)
        ctx.nll = nll
        return ops.reduce_sum(nll, 1.0 / B)

    @staticmethod
    def backward(ctx, g):
        e, d, labels, tlens, ulens, ws = ctx.saved_tensors
        w_out, b_out = ctx.pr
        blank, act, plan = ctx.cfg
        B, T, J = e.shape
        U = d.shape[1]
        adt = ops.act_dtype()
        sink = GradSink(ctx.pr)
        gs = g.reshape(1).contiguous()
        de = torch.zeros_like(e)
        dd = torch.zeros_like(d)
        for b, t0, nt in plan:
            eb, db_ = e[b:b + 1, t0:t0 + nt], d[b:b + 1]
            H = ops.joint_fwd(eb, db_, act, out_dtype=adt)
            H2 = H.view(nt * U, J)
            # the recomputed logits leave the GEMM as their gradient (epilogue 8); declined shapes: logits, then a gradient pass
            dZ = ops.rnnt_node_grad_fused(H2, ops.wshadow(w_out), b_out, labels, tlens, ulens, ws, (b * T + t0) * U, B, T, U,
                                          blank, gs, 1.0 / B, adt)
            if dZ is None:
                Z = ops.linear_fwd(H2, ops.wshadow(w_out), b_out)
                dZ = ops.rnnt_node_grad(Z, labels, tlens, ulens, ws, (b * T + t0) * U, B, T, U, blank, gs, 1.0 / B, out_dtype=adt)
            ops.linear_bwd_w(dZ, H2, sink.buf(0), db=sink.buf(1), group=False)   # per-chunk: queueing would keep every chunk alive
            dH = ops.linear_bwd_x(dZ, ops.wshadow(w_out))
            de_c, dd_c = ops.joint_bwd(dH.view(1, nt, U, J), eb.contiguous(), db_, act)
            de[b, t0:t0 + nt].copy_(de_c[0])
            ops.axpby(dd[b], dd_c[0], 1.0, 1.0, out=dd[b])
        return (de, dd) + sink.results() + (None,) * 7
