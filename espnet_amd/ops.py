"""Thin tensor-level wrappers over the C ABI (one Python function per kernel family).

No arithmetic happens here: every function validates shapes on the host (a faulting kernel can
take the whole GPU node down), fills the argument struct and launches on torch's current stream.
"""
import ctypes as C
import os
import math

import torch

from . import _lib
from ._lib import GatherT, GemmT, RowMapT, check, ptr, stream_ptr

ACT_NONE, ACT_RELU, ACT_SWISH = 0, 1, 2
EPI_NONE, EPI_RELU, EPI_SWISH, EPI_MUL_RELU_MASK, EPI_MUL_DSWISH, EPI_MUL_AUX, EPI_DACT_FACTOR, EPI_ROW_STATS, EPI_ROW_GRAD = 0, 1, 2, 3, 4, 5, 6, 7, 8

_PRECISIONS = {"fp32": 0, "bf16": 1}
_state = {"precision": 0}


def set_precision(name):
    """'fp32' -> v_mfma_f32_16x16x4_f32 (exact fp32 products, parity mode);
    'bf16' -> v_mfma_f32_16x16x32_bf16 (operands rounded to bf16 on load, fp32 accumulate)."""
    if name not in _PRECISIONS:
        raise ValueError(f"precision must be one of {list(_PRECISIONS)}")
    _state["precision"] = _PRECISIONS[name]


def get_precision():
    return [k for k, v in _PRECISIONS.items() if v == _state["precision"]][0]


def fast():
    """bf16 mode: GEMM operands (activations + weight shadows) live in HBM as bf16."""
    return _state["precision"] == 1


def f32_epilogue_drop():
    """fp32 mode: block-output dropout + residual ride in the epilogue of the pipelined fp32 GEMM (gemm_f32.hip); the
    operands of every block GEMM are 16-byte aligned with K % 4 == 0, which is all that kernel asks for"""
    return _state["precision"] == 0 and F32_EPILOGUE_DROP


F32_EPILOGUE_DROP = True     # tests flip this to compare against the stand-alone dropout kernel


# Off by default: measured on configs[1] the staging-side hash + activation pushes the 128x128 instantiations over the
# 256-VGPR budget (scratch) and the 64x64 ones from 3 to 2 waves per SIMD - the GEMMs lose 2.3 ms per step where the
# stand-alone dropout passes they replace cost 1.1 ms.  Kept (and tested) as an operator feature.
F32_OPERAND_DROP = False


def f32_operand_drop():
    """fp32 mode: dropout of a GEMM operand (FFN inner dropout, every block's incoming-gradient dropout) applied while
    the operand is staged, instead of a pass that materialises the dropped tensor"""
    return _state["precision"] == 0 and F32_OPERAND_DROP


def act_dtype():
    return torch.bfloat16 if fast() else torch.float32


def wshadow(p):
    """bf16 shadow of a parameter (fast mode) or the parameter itself (fp32 mode).
    FlatParams keeps an arena shadow that the Adam kernel refreshes; otherwise (or after an external
    in-place modification, detected through tensor._version) the shadow is re-cast on demand."""
    if not fast():
        return p
    sh = getattr(p, "_eamd_bf16", None)
    if sh is not None and getattr(p, "_eamd_bf16_ver", None) == p._version:
        return sh
    if sh is None:
        sh = torch.empty(p.shape, device=p.device, dtype=torch.bfloat16)
        p._eamd_bf16 = sh
    cast_bf16(p.detach().contiguous(), sh)
    p._eamd_bf16_ver = p._version
    return sh


def to_act(x):
    """fp32 activation -> GEMM operand dtype of the current mode"""
    if fast() and x.dtype == torch.float32:
        return cast_bf16(x.contiguous())
    return x


def to_act_shared(x):
    """to_act for a tensor that several consumers cast within one step (the encoder memory read by every decoder
    layer, the positional embedding read by every encoder layer): the bf16 copy rides on the tensor object"""
    if not (fast() and x.dtype == torch.float32):
        return x
    c = getattr(x, "_eamd_act", None)
    if c is not None and c[1] == x._version and c[0].shape == x.shape:
        return c[0]
    y = cast_bf16(x.contiguous())
    x._eamd_act = (y, x._version)
    return y


def _numel_from(t, off):
    return t.numel() - off


def _span(rows, ld, cols):
    """elements touched by a [rows, cols] matrix with leading dimension ld"""
    return (rows - 1) * ld + cols if rows > 0 else 0


# ---- time bound of a padded batch ---------------------------------------------------------------------------------
# A shape-bucketed graph (train.BucketedGraphStep) pads a batch's frame axis beyond its own longest utterance.  The
# reference never sees those frames, and three places of the Conformer encoder depend on the batch's own length: the
# legacy rel_shift (attention.py:160-171, taken over T' x T'), the depthwise convolution's zero padding and the BatchNorm
# statistics (convolution.py:56-79, over B x T' frames).  While a bound is set (device int32 scalar = the batch's own
# subsampled length, read by the kernels at run time, i.e. per graph replay) those three follow it.  The blocks read the
# bound in their FORWARD and keep it for their backward; the model clears it at the end of its forward pass.
_time_bound = {"t": None}


def set_time_bound(t):
    """t: int32 device tensor with one element, or None"""
    if t is not None and not (t.is_cuda and t.dtype == torch.int32 and t.numel() == 1):
        raise _lib.EamdError("time bound: one int32 element on the device")
    _time_bound["t"] = t


def time_bound():
    return _time_bound["t"]


def mask_time(x, T, bound):
    """rows (b, t) of the [B*T, C] tensor x with t >= bound[0] <- 0, in place"""
    rows = x.numel() // x.shape[-1]
    assert x.is_contiguous() and rows % T == 0 and x.dtype in (torch.float32, torch.bfloat16)
    check(_lib.lib().eamd_mask_time(ptr(x), C.c_int64(rows), x.shape[-1], T, ptr(bound), int(x.dtype == torch.bfloat16),
                                    stream_ptr()), "eamd_mask_time")
    return x


# bench.py: list collecting every MFMA-contraction launch of one step as (descriptor or None, operand references,
# replay(stream_ptr)) - eamd_gemm descriptors and the fused attention kernels (descriptor None)
_gemm_record = None


def gemm(A, B, Cm, M, N, K, lda, ldb, ldc, *, transA=0, transB=0, bias=None, aux=None, ldaux=0, R=None,
         ldr=0, batch=(1, 1), sA=(0, 0), sB=(0, 0), sC=(0, 0), alpha=1.0, beta=0.0, a_act=0, b_act=0,
         epilogue=0, splitk=1, tile=0, gather=None, cmap=None, a_off=0, b_off=0, c_off=0, precision=None,
         colsum=None, Cb=None, drop=None, Hb=None, h_act=0, a_drop=None, b_drop=None, group=False, stats=None, defer=None):
    """A, B: both float32 or both bfloat16 (bf16 operands select the fast MFMA kernel).
    Cm: float32 result, or bfloat16 result (then no fp32 copy is written); Cb: extra bf16 copy.
    defer=list: the checked descriptor is appended to the list instead of being launched; gemm_multi(list) then issues
    all of them together (eamd_gemm_multi: independent products, e.g. the parity classes of a strided convolution's dX).
    group=True (weight gradients): while wgrad_group_begin() is in effect the launch is queued and leaves with all other
    queued ones as ONE grouped launch at wgrad_join() (eamd_gemm_group_*), if the library accepts it for that."""
    bf = A.dtype == torch.bfloat16
    if A.dtype != B.dtype or A.dtype not in (torch.float32, torch.bfloat16) or not (A.is_cuda and B.is_cuda):
        raise _lib.EamdError("gemm needs float32 or bfloat16 GPU operands of one dtype")
    if stats is not None:      # EPI_ROW_STATS: (part, col, zcol, zfix, fix), Cm = None;  EPI_ROW_GRAD: (rowc, col, fix, gscale, scale)
        assert epilogue in (EPI_ROW_STATS, EPI_ROW_GRAD) and tile in (64, 128) and splitk == 1
        assert (Cm is None and Cb is None) if epilogue == EPI_ROW_STATS else Cm is not None
    c32 = Cm if (Cm is not None and Cm.dtype == torch.float32) else None
    c16 = Cb if Cb is not None else (Cm if (Cm is not None and Cm.dtype == torch.bfloat16) else None)
    if c16 is not None and (c16.dtype != torch.bfloat16 or not bf):
        raise _lib.EamdError("gemm: bf16 outputs need bf16 operands")
    for tt in (R, bias, colsum):
        if tt is not None and tt.dtype != torch.float32:
            raise _lib.EamdError("gemm: bias / residual / colsum must be float32")
    b1, b2 = batch
    # ---- host-side bounds checks -------------------------------------------------------------
    if gather is None:
        a_span = _span(K, lda, M) if transA else _span(M, lda, K)
        a_need = a_off + (b1 - 1) * sA[0] + (b2 - 1) * sA[1] + a_span
        if a_need > A.numel():
            raise _lib.EamdError(f"gemm: A too small ({A.numel()} < {a_need})")
    b_span = _span(K, ldb, N) if transB else _span(N, ldb, K)
    b_need = b_off + (b1 - 1) * sB[0] + (b2 - 1) * sB[1] + b_span
    if b_need > B.numel():
        raise _lib.EamdError(f"gemm: B too small ({B.numel()} < {b_need})")
    if cmap is None:
        c_need = c_off + (b1 - 1) * sC[0] + (b2 - 1) * sC[1] + _span(M, ldc, N)
        for ct in (c32, c16):
            if ct is not None and c_need > ct.numel():
                raise _lib.EamdError(f"gemm: C too small ({ct.numel()} < {c_need})")
    if bias is not None and bias.numel() < N:
        raise _lib.EamdError("gemm: bias too small")

    p = GemmT()
    p.A, p.B = ptr(A, a_off), ptr(B, b_off)
    p.C = ptr(c32, c_off) if c32 is not None else None
    p.Cb = ptr(c16, c_off) if c16 is not None else None
    p.bias = ptr(bias)
    p.aux = ptr(aux, c_off) if aux is not None else None
    p.aux_dtype = 1 if (aux is not None and aux.dtype == torch.bfloat16) else 0
    p.R = ptr(R, c_off) if R is not None else None
    if colsum is not None:
        if not transA or gather is not None or colsum.numel() < M * b1 * b2:
            raise _lib.EamdError("gemm: colsum needs transA, no gather and M outputs")
        p.colsum = ptr(colsum)
    p.M, p.N, p.K = M, N, K
    p.transA, p.transB = int(transA), int(transB)
    p.lda, p.ldb, p.ldc, p.ldaux, p.ldr = lda, ldb, ldc, ldaux or ldc, ldr or ldc
    p.batch1, p.batch2 = b1, b2
    p.sA1, p.sA2 = sA
    p.sB1, p.sB2 = sB
    p.sC1, p.sC2 = sC
    p.alpha, p.beta = alpha, beta
    p.a_act, p.b_act, p.epilogue = a_act, b_act, epilogue
    p.splitk = splitk
    p.in_dtype = 1 if bf else 0
    p.precision = 1 if bf else (_state["precision"] if precision is None else precision)
    p.tile = tile
    if stats is not None and epilogue == EPI_ROW_GRAD:
        rowc, col, fix, gscale, scale = stats
        if rowc.dtype != torch.float32 or rowc.numel() < 3 * M or col.dtype != torch.int32 or col.numel() < M or not (0 <= fix < N):
            raise _lib.EamdError("gemm: row-gradient coefficients too small or of the wrong dtype")
        p.stats.rowc, p.stats.col, p.stats.fix, p.stats.gscale, p.stats.scale = ptr(rowc), ptr(col), int(fix), ptr(gscale), float(scale)
    elif stats is not None:
        part, col, zcol, zfix, fix = stats
        tn = (N + tile - 1) // tile
        if part.dtype != torch.float32 or part.numel() < M * tn * 2 or zfix.numel() < M or not (0 <= fix < N) or \
                (col is not None and (col.dtype != torch.int32 or col.numel() < M or zcol is None or zcol.numel() < M)):
            raise _lib.EamdError("gemm: row-statistics buffers too small or of the wrong dtype")
        p.stats.part, p.stats.col, p.stats.zcol, p.stats.zfix, p.stats.fix = ptr(part), ptr(col), ptr(zcol), ptr(zfix), int(fix)
    if gather is not None:
        p.gather = gather
    if cmap is not None:
        p.cmap = cmap
    if drop is not None and drop[0] > 0.0:      # (p, salt): fused dropout, see include/espnet_amd.h
        if ldc != N or c_off != 0 or (not bf and p.precision != 0):
            raise _lib.EamdError("gemm: fused dropout needs a contiguous [M, N] result and bf16 or fp32-MFMA operands")
        p.drop_p, p.drop_salt = float(drop[0]), int(drop[1])
        p.drop_step = ptr(rng_state(A.device))
        if Hb is not None:
            if Hb.dtype != (torch.bfloat16 if bf else torch.float32) or Hb.numel() < M * N:
                raise _lib.EamdError("gemm: Hb must be an [M, N] buffer of the operand dtype")
            p.Hb, p.h_act, p.h_dtype = ptr(Hb), h_act, 0 if bf else 1
    for which, d, ld_, off_, cols in (("a", a_drop, lda, a_off, (M if transA else K)), ("b", b_drop, ldb, b_off, (N if transB else K))):
        if d is not None and d[0] > 0.0:    # operand-side dropout: the operand must BE the contiguous tensor the mask was drawn for
            if bf or p.precision != 0 or ld_ != cols or off_ != 0 or b1 * b2 != 1 or gather is not None:
                raise _lib.EamdError("gemm: operand dropout needs an fp32-MFMA launch on a dense, unbatched operand")
            setattr(p, which + "_drop_p", float(d[0]))
            setattr(p, which + "_drop_salt", int(d[1]))
            p.drop_step = ptr(rng_state(A.device))
    if group and _wgroup["on"] and _in_grad_arena(Cm) and _wgroup_accepts(p):
        # Two queued problems that accumulate into the same dW may share one launch only if BOTH add with atomics:
        # splitk == 1 with beta == 1 is a plain read-modify-write of the tile, and workgroups of one launch are not
        # ordered (a weight used twice per backward pass with 65..511 reduction rows: per-step decoder cells, shared
        # parameters).  The earlier problems leave first (stream order keeps the sums exact); so does the queue when
        # the operands it keeps alive exceed GROUP_WGRAD_MAX_BYTES (streamed losses queue one weight gradient per chunk).
        lo = Cm.data_ptr() + 4 * c_off
        hi = lo + 4 * _span(M, ldc, N)
        nbytes = A.numel() * A.element_size() + B.numel() * B.element_size()
        clash = any(lo < qhi and qlo < hi and (splitk == 1 or qsk == 1) for qlo, qhi, qsk in _wgroup["ranges"])
        if clash or _wgroup["bytes"] + nbytes > GROUP_WGRAD_MAX_BYTES:
            wgrad_group_flush()
        _wgroup["items"].append((p, (A, B, Cm, colsum)))
        _wgroup["ranges"].append((lo, hi, splitk))
        _wgroup["bytes"] += nbytes
        return
    if defer is not None:
        defer.append((p, (A, B, Cm, bias, aux, R, colsum, Cb, Hb, stats)))
        return
    if _gemm_record is not None:
        _gemm_record.append((p, (A, B, Cm, bias, aux, R, colsum, Cb, Hb, stats),
                             lambda sp, p=p: check(_lib.lib().eamd_gemm(C.byref(p), sp), "eamd_gemm")))
    check(_lib.lib().eamd_gemm(C.byref(p), stream_ptr()), "eamd_gemm")


GEMM_MULTI_MAX = 4


def gemm_multi(items):
    """issue the products collected with gemm(..., defer=items) together: up to GEMM_MULTI_MAX per eamd_gemm_multi call (one
    launch where the library has a kernel for the combination, otherwise one launch each, in order)"""
    L = _lib.lib()
    for i0 in range(0, len(items), GEMM_MULTI_MAX):
        part = items[i0:i0 + GEMM_MULTI_MAX]
        arr = (GemmT * len(part))(*[it[0] for it in part])
        n = len(part)
        if _gemm_record is not None and any(it[0].in_dtype != 0 or it[0].precision != 0 for it in part):
            for p_, keep in part:          # no one-launch kernel for these operands: n launches, recorded as such
                _gemm_record.append((p_, keep, lambda sp, p_=p_: check(L.eamd_gemm(C.byref(p_), sp), "eamd_gemm")))
        elif _gemm_record is not None:
            _gemm_record.append((dict(kind="multi%d" % n, flop=sum(2.0 * it[0].M * it[0].N * it[0].K for it in part)),
                                 (arr, [it[1] for it in part]),
                                 lambda sp, arr=arr, n=n: check(L.eamd_gemm_multi(arr, n, sp), "eamd_gemm_multi")))
        check(L.eamd_gemm_multi(arr, n, stream_ptr()), "eamd_gemm_multi")


# ---- grouped weight-gradient launches ------------------------------------------------------------
# dW = dY^T X of a small layer (256 x 256 ... 768 x 256 outputs) is 16-48 tiles: even with split-K such a launch runs
# at a quarter of what the chip can do and costs a launch slot.  Nothing in backward reads dW, so between
# wgrad_group_begin() and wgrad_join() every eligible one is queued (operands kept alive) and all of them leave as ONE
# launch whose workgroups look their problem up in a device table (eamd_gemm_group_plan / eamd_gemm_group_launch).
GROUP_WGRAD = True            # tests flip this to reach the one-launch-per-GEMM path
STACK_WGRAD_MAX_ROWS = 64     # weight gradients of at most this many rows are stacked along the reduction until the flush (0: off)
GROUP_WGRAD_MAX_TILES = int(os.environ.get("EAMD_GROUP_MAX_TILES", "384"))    # 64x64 output tiles: larger weight gradients fill the chip on their own
# measured at config 2 (tools/group_sweep.sh): grouping everything up to 384 output tiles (all but the vocabulary-sized
# gradients) with half the stand-alone split count is best in both precisions - 34.09 ms fp32 / 14.20 ms bf16 against
# 35.0 / 15.04 ungrouped; the queue supplies the parallelism the extra splits (and their atomics) bought before
GROUP_WGRAD_SK_DIV = int(os.environ.get("EAMD_GROUP_SK_DIV", "2"))
# fp32 operands only (measured: 33.45 -> 32.80 ms; with bf16 operands the 128x128 tile loses, 14.17 -> 14.3+ ms): problems
# of at least this many 64x64 tiles - the FFN weights - take 128x128 tiles in a second grouped launch, with about
# GROUP_WGRAD_T128_WGS workgroups (tiles x K-splits) each
GROUP_WGRAD_TILE128_MIN = int(os.environ.get("EAMD_GROUP_T128_MIN", "100"))
GROUP_WGRAD_T128_WGS = int(os.environ.get("EAMD_GROUP_T128_WGS", "96"))
# operand bytes the queue may keep alive before it is flushed early (the operands of every queued problem stay
# allocated until the grouped launch has been issued)
GROUP_WGRAD_MAX_BYTES = int(os.environ.get("EAMD_GROUP_MAX_BYTES", str(16 << 30)))     # config 2 queues 4.6 GB per backward pass
_wgroup = {"on": False, "items": [], "ranges": [], "bytes": 0, "stack": {}, "pinned": [], "reserve": [], "arenas": {}}


def register_grad_arena(t):
    """only GEMMs that accumulate into a registered gradient arena (espnet_amd.train.FlatParams.grad) are queued: a
    temporary result (tap-major weight gradients that are permuted into the arena next, gradients handed back to
    autograd) is read by its consumer long before the grouped launch runs"""
    import weakref
    key = t.untyped_storage().data_ptr()
    _wgroup["arenas"][key] = weakref.ref(t)


def _in_grad_arena(t):
    """True if `t` lives in a registered gradient arena that is still alive (a freed arena's address may be handed to
    any later allocation)"""
    key = t.untyped_storage().data_ptr()
    ref = _wgroup["arenas"].get(key)
    if ref is None:
        return False
    a = ref()
    if a is None or a.untyped_storage().data_ptr() != key:
        del _wgroup["arenas"][key]
        return False
    return True


def wgrad_group_begin():
    _wgroup["on"] = GROUP_WGRAD
    _wgroup["items"], _wgroup["stack"] = [], {}      # nothing queued by a backward pass that ended in an exception survives
    _wgroup["ranges"], _wgroup["bytes"] = [], 0


def _wgroup_accepts(p):
    if ((p.M + 63) // 64) * ((p.N + 63) // 64) > GROUP_WGRAD_MAX_TILES:
        return False
    first = (C.c_int32 * 2)()
    return _lib.lib().eamd_gemm_group_plan(C.byref(p), 1, first) > 0


def _wgroup_staging(nbytes):
    """pinned host buffer for one descriptor table.  A captured graph copies from its buffer again at every replay, so
    a capture takes a buffer for good out of a small reserve that eager flushes (every capture is preceded by eager
    warm-up steps) keep topped up - pinned memory cannot be allocated while a stream is capturing."""
    if torch.cuda.is_current_stream_capturing():
        for i, h in enumerate(_wgroup["reserve"]):
            if h.numel() >= nbytes:
                _wgroup["pinned"].append(_wgroup["reserve"].pop(i))
                return _wgroup["pinned"][-1][:nbytes]
        return None
    want = (nbytes + 4095) // 4096 * 4096 * 2
    _wgroup["reserve"] = [h for h in _wgroup["reserve"] if h.numel() >= want]
    while len(_wgroup["reserve"]) < 16:
        _wgroup["reserve"].append(torch.empty(want, dtype=torch.uint8, pin_memory=True))
    return torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)


# EAMD_WGRAD_GROUP_SIDE=1: the grouped launches leave on a SECOND stream (the backward pass's input-gradient chain goes on beside
# them; their operands are kept alive until wgrad_join) - with EAMD_GROUP_MAX_BYTES set so that a backward pass flushes several
# times.  An experiment knob: see DESIGN.md for what it measured.
WGROUP_SIDE = os.environ.get("EAMD_WGRAD_GROUP_SIDE", "0") == "1"


def wgrad_group_flush():
    """launch everything queued so far: one grouped launch per tile size (a single queued GEMM goes out on its own)"""
    if WGROUP_SIDE and (_wgroup["items"] or _wgroup["stack"]):
        cur = torch.cuda.current_stream()
        side = _wgroup.get("side")
        if side is None or side.device != cur.device:
            side = _wgroup["side"] = torch.cuda.Stream(device=cur.device)
        _wgroup.setdefault("held", []).append(([it[1] for it in _wgroup["items"]], list(_wgroup["stack"].values())))
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            _wgroup_flush_now()
        _wgroup["side_used"] = True
        return
    _wgroup_flush_now()


def _wgroup_flush_now():
    stack, _wgroup["stack"] = _wgroup["stack"], {}
    for dW, db, dys, xs in stack.values():          # stacked small-M weight gradients: one product per weight
        if len(dys) == 1:
            dyc, xc = dys[0], xs[0]
        else:
            dyc, xc = torch.cat(dys, 0), torch.cat(xs, 0)
        if dyc.shape[0] <= STACK_WGRAD_MAX_ROWS:    # still a handful of rows: straight to the GEMM
            sk = auto_splitk(dyc.shape[1], xc.shape[1], dyc.shape[0])
            gemm(dyc, xc, dW, dyc.shape[1], xc.shape[1], dyc.shape[0], dyc.shape[1], xc.shape[1], xc.shape[1], transA=1,
                 transB=1, splitk=sk, beta=1.0 if sk == 1 else 0.0, colsum=db, group=True)
        else:
            _linear_bwd_w(dyc, xc, dW, db=db)
    items = _wgroup["items"]
    if not items:
        return
    _wgroup["items"], _wgroup["ranges"], _wgroup["bytes"] = [], [], 0
    for tile in (64, 128):
        part = [it for it in items if (128 if it[0].tile == 128 else 64) == tile]
        if part:
            _wgroup_launch(part, tile)


def _wgroup_launch(items, tile):
    L = _lib.lib()

    def separately():
        for p, keep in items:
            if _gemm_record is not None:
                _gemm_record.append((p, keep, lambda sp, p=p: check(L.eamd_gemm(C.byref(p), sp), "eamd_gemm")))
            check(L.eamd_gemm(C.byref(p), stream_ptr()), "eamd_gemm")

    if len(items) == 1:
        return separately()
    n = len(items)
    dev = items[0][1][0].device
    arr = (GemmT * n)(*[it[0] for it in items])
    first = (C.c_int32 * (n + 1))()
    total = L.eamd_gemm_group_plan(arr, n, first)
    check(min(total, 0), "eamd_gemm_group_plan")
    nb_desc, nb_first = C.sizeof(arr), C.sizeof(first)
    host = _wgroup_staging(nb_desc + nb_first)
    if host is None:          # capturing with no staging buffer in reserve: one launch per GEMM
        return separately()
    C.memmove(host.data_ptr(), arr, nb_desc)
    C.memmove(host.data_ptr() + nb_desc, first, nb_first)
    tab = torch.empty(nb_desc + nb_first, dtype=torch.uint8, device=dev)
    tab.copy_(host, non_blocking=True)
    in_dtype = int(arr[0].in_dtype)
    args = (ptr(tab), ptr(tab, nb_desc), n, total, in_dtype, tile)
    check(L.eamd_gemm_group_launch(*args, stream_ptr()), "eamd_gemm_group_launch")
    if _gemm_record is not None:
        _gemm_record.append((dict(kind="group%d" % tile, flop=sum(2.0 * it[0].M * it[0].N * it[0].K * it[0].batch1 * it[0].batch2 for it in items)),
                             (tab, [it[1] for it in items]),
                             lambda sp, args=args: check(L.eamd_gemm_group_launch(*args, sp), "eamd_gemm_group_launch")))


def cast_bf16(x, out=None):
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16)
    assert x.dtype == torch.float32 and x.is_contiguous() and out.numel() == x.numel()
    check(_lib.lib().eamd_cast_bf16(ptr(x), ptr(out), C.c_int64(x.numel()), stream_ptr()), "eamd_cast_bf16")
    return out


def auto_splitk(m_out, n_out, k_red):
    """split count for reduction-heavy GEMMs (dW = dY^T X).  Measured on MI355X (tools/gemm_probe2.py,
    K = 7968): 128 output tiles -> 4, 32 tiles -> 8, 16 tiles -> 16 are the fastest; more splits lose to
    the f32 atomics, fewer leave CUs idle.  Powers of two >= 8 keep whole K-slices per XCD (gemm_bf16.hip:
    split index = workgroup id mod splitk)."""
    tiles = ((m_out + 63) // 64) * ((n_out + 63) // 64)
    if tiles >= 512:
        return 1
    want = 4.0 * (128.0 / tiles) ** 0.5
    s = 1
    while s < want:
        s *= 2
    return max(1, min(s, 64, max(1, k_red // 128)))


# ---- nn.Linear pieces -------------------------------------------------------------------------
LINEAR_ROWS = os.environ.get("EAMD_LINEAR_ROWS", "1") != "0"      # eamd_linear_rows_f32 for few rows without autograd
_infer = 0      # > 0 inside an explicit inference region (ops.inference / @ops.inference_call: searches, F_.run's direct forward)


class inference:
    """marks an inference region: products of a handful of rows take eamd_linear_rows_f32 there.  NOT keyed on
    torch.is_grad_enabled(): grad mode is also off inside every autograd.Function.forward of a TRAINING step, where the few-row
    products (a small batch through the RNN decoder) must keep the summation order of the GEMMs used for their gradients."""

    def __enter__(self):
        global _infer
        _infer += 1

    def __exit__(self, *exc):
        global _infer
        _infer -= 1
        return False


def inference_call(fn):
    import functools

    @functools.wraps(fn)
    def wrapped(*a, **k):
        with inference():
            return fn(*a, **k)
    return wrapped
# rows up to which it is used.  Beyond 16 rows the library has a one-wave-per-16x16-tile MFMA form (direct global loads, no LDS).
# Measured in the replayed step of a batched beam search (M = 320 rows, tools/trace_decode_graph.sh, round 4): 7.0 us against
# 13.1 us of the 64 x 64 tiles for N = 256 / 768 at K = 256 (those are 20 / 60 workgroups walking K alone), the same for N = 2048,
# SLOWER for K = 2048 (33 against 26 us with split-K: every wave re-reads its 32 rows of both operands) and for the output
# layer (N = 5000: 29.7 against 17.8 us) - so it takes the narrow products only
LINEAR_ROWS_MAX = int(os.environ.get("EAMD_LINEAR_ROWS_MAX", "16"))
LINEAR_ROWS_BLOCK_MAX = int(os.environ.get("EAMD_LINEAR_ROWS_BLOCK_MAX", "1024"))      # ... up to this many rows, K <= 512, N <= 1024
# ... and the long reductions (K >= 1024: the waves of a workgroup split K, one 16 x 16 tile per workgroup) - A/B knob
LINEAR_ROWS_KSPLIT = os.environ.get("EAMD_LINEAR_ROWS_KSPLIT", "1") != "0"


def linear_fwd(x, W, b, out=None, *, act=EPI_NONE, R=None, alpha=1.0, a_act=ACT_NONE, out_dtype=torch.float32,
               drop=None, Hb=None, h_act=ACT_NONE, a_drop=None):
    """out[M,N] = alpha * drop(act(a_act(x)[M,K] @ W[N,K]^T + b)) + R      (x, W: both fp32 or both bf16)
    drop = (p, salt): dropout fused in the epilogue (bf16 operands); with Hb the value itself is left
    alone and Hb <- dropout(h_act(value)) is written as a second bf16 output."""
    M, K = x.shape
    N = W.shape[0]
    assert W.shape[1] == K and W.is_contiguous() and x.stride(1) == 1 and (R is None or R.stride(-1) == 1)
    if (LINEAR_ROWS and (M <= LINEAR_ROWS_MAX or (M <= LINEAR_ROWS_BLOCK_MAX and (K <= 512 or (K >= 1024 and LINEAR_ROWS_KSPLIT)) and N <= 1024
                                                  and K % 16 == 0)) and out is None and out_dtype == torch.float32 and x.dtype == torch.float32 and W.dtype == torch.float32
            and act in (EPI_NONE, EPI_RELU, EPI_SWISH) and a_act in (ACT_NONE, ACT_RELU, ACT_SWISH) and drop is None and Hb is None
            and a_drop is None and K % 4 == 0 and _state["precision"] == 0 and _infer > 0 and not torch.is_grad_enabled()
            and (R is None or (tuple(R.shape) == (M, N) and (M == 1 or R.stride(0) >= N))) and (M == 1 or x.stride(0) >= K)):
        # (an expanded, stride-0 operand with M > 1 would read as `dense` on the C side: those go to the GEMM below via .contiguous())
        # inference on a handful of rows (one utterance's hypotheses in a beam step): one wave per output column; x and R may be
        # row-strided views (the newest position of every prefix)
        y = torch.empty(M, N, device=x.device, dtype=torch.float32)
        rc = _lib.lib().eamd_linear_rows_f32(ptr(x), ptr(W), ptr(b), ptr(R), ptr(y), M, N, K, a_act, act, C.c_float(alpha),
                                             C.c_int64(x.stride(0)), C.c_int64(R.stride(0) if R is not None else 0), stream_ptr())
        if rc == 0:
            return y
        if rc != _lib.EAMD_EUNSUPPORTED:
            check(rc, "eamd_linear_rows_f32")
    x = x.contiguous()
    R = R.contiguous() if R is not None else None
    sk = 1
    if (out_dtype if out is None else out.dtype) == torch.float32 and act == EPI_NONE and drop is None and Hb is None and a_drop is None:
        if R is None and alpha == 1.0:
            sk = _skinny_splitk(M, N, K)
        elif K >= 1024 and M <= 512 and x.is_cuda and torch.cuda.is_current_stream_capturing():
            # a beam-search step being captured: the second feed-forward product of a few (hundred) hypotheses walks K = 2048 in
            # 4 .. 20 workgroups, 58 us; eagerly the extra zero-fill launch costs the host more than the device saves, in a graph it
            # does not.  Bias, alpha and the residual come with the first K slice.
            sk = _skinny_splitk(M, N, K, max_rows=512)
    if out is None:
        out = zeros(M, N, device=x.device) if sk > 1 else torch.empty(M, N, device=x.device, dtype=out_dtype)
    elif sk > 1:
        out.zero_()
    gemm(x, W, out, M, N, K, K, K, N, bias=b, epilogue=act, R=R, ldr=N, alpha=alpha, a_act=a_act, drop=drop, Hb=Hb,
         h_act=h_act, a_drop=a_drop, splitk=sk)
    return out


# ---- zero arena --------------------------------------------------------------------------------------------------
# A decoder loop asks for hundreds of small zero-filled buffers per training step (split-K results that accumulate
# through atomics, gradient accumulators of the attention steps): 716 fill launches of 4 us at BASELINE config 4.
# Between zero_arena_begin() (the model's forward in training mode) and the next one, zeros() hands out fresh
# slices of ONE buffer that a single fill zeroed at begin; its size is the previous step's demand (the first step,
# and anything beyond the buffer, falls back to torch.zeros).  Slices are never handed out twice.
class _ZeroArena:
    buf = None
    off = 0
    need = 0        # bytes asked for since the last begin, counted only between a begin and the next begin / off
    cap = 0         # bytes the next arena gets
    active = False
    tracking = False
    capture = 0     # id of the stream capture zero_arena_begin ran in (0 = eager)


_zarena = _ZeroArena()
ZERO_ARENA = os.environ.get("EAMD_ZERO_ARENA", "1") != "0"
ZERO_ARENA_MAX = int(os.environ.get("EAMD_ZERO_ARENA_MAX_MB", "1024")) << 20     # larger demands go to torch.zeros


def _capture_id():
    """0 when the current stream is not capturing, else a number that tells captures apart (the buffer an arena hands out
    was zeroed by a fill INSIDE one capture, or eagerly: a slice may only be used where that fill runs too)"""
    if not torch.cuda.is_available() or not torch.cuda.is_current_stream_capturing():
        return 0
    try:
        from . import graphs
        return graphs.capture_id()
    except Exception:  # noqa: BLE001 - no id available: "some capture"
        return -1


def zero_arena_begin(device):
    """a FRESH buffer per step (one allocation + one fill): slices a still-living autograd graph of an earlier forward
    holds keep their storage alive, so a second forward before that backward cannot clobber them.  Its size follows the
    demand of the step that just ended (up AND down), capped at ZERO_ARENA_MAX."""
    a = _zarena
    if a.tracking:
        a.cap = min(ZERO_ARENA_MAX, (int(a.need * 1.25) + 4095) // 4096 * 4096)
    a.off, a.need, a.tracking = 0, 0, True
    a.active = ZERO_ARENA and a.cap > 0
    a.capture = _capture_id() if (a.active and torch.device(device).type == "cuda") else 0
    a.buf = torch.zeros(a.cap // 4, device=device, dtype=torch.float32) if a.active else None


def zero_arena_off():
    """inference paths and everything else outside a training forward / backward: zeros() is torch.zeros again and its
    demand is not counted towards the next training step's arena"""
    a = _zarena
    a.active, a.tracking, a.buf, a.off, a.need, a.capture = False, False, None, 0, 0, 0


def zeros(*shape, device):
    """fp32 zeros: a slice of the step's zero arena when one is active, else torch.zeros"""
    if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
        shape = tuple(shape[0])
    n = 1
    for d in shape:
        n *= int(d)
    a = _zarena
    nb = (n * 4 + 255) // 256 * 256
    if a.tracking:
        a.need += nb
    if a.active and n > 0 and a.off + nb <= a.buf.numel() * 4 and a.buf.device == torch.device(device) \
            and (not a.buf.is_cuda or _capture_id() == a.capture):      # never a slice zeroed outside the capture that uses it
        t = a.buf[a.off // 4: a.off // 4 + n].view(shape)
        a.off += nb
        return t
    return torch.zeros(shape, device=device, dtype=torch.float32)


def _skinny_splitk(M, N, K, max_rows=64):
    """a product with a handful of rows (one decoder step: M = batch, or batch x beam) is one row of 64-wide tiles - 16 to
    64 workgroups, each walking the whole reduction alone while the weight matrix streams through a fraction of the
    chip: split the reduction until ~256 workgroups share it (f32 atomics into the zeroed result; config 4's decoder
    steps: 20 -> 9 us for the 32 x 2048 x 4096 input gradient)"""
    if M > max_rows or N * K < 512 * 512:       # small weights: nothing to stream, keep the single-pass summation order
        return 1
    tiles = (M + 63) // 64 * ((N + 63) // 64)
    sk = 1
    while tiles * sk < 256 and K // (sk * 2) >= 128:
        sk *= 2
    return sk


def linear_bwd_x(dy, W, out=None, *, beta=0.0, epilogue=EPI_NONE, aux=None, alpha=1.0, out_dtype=torch.float32,
                 drop=None, a_drop=None):
    """out[M,K] = alpha * drop(epi(dy[M,N] @ W[N,K])) + beta*out"""
    M, N = dy.shape
    K = W.shape[1]
    assert W.shape[0] == N and dy.is_contiguous()
    fresh = out is None
    sk = 1
    if fresh and out_dtype == torch.float32 and epilogue == EPI_NONE and aux is None and drop is None and a_drop is None \
            and alpha == 1.0:
        sk = _skinny_splitk(M, K, N)
    if out is None:
        assert beta == 0.0
        out = zeros(M, K, device=dy.device) if sk > 1 else torch.empty(M, K, device=dy.device, dtype=out_dtype)
    gemm(dy, W, out, M, K, N, N, K, K, transB=1, beta=beta, epilogue=epilogue, aux=aux, ldaux=K, alpha=alpha,
         drop=drop, a_drop=a_drop, splitk=sk)
    return out


# ---- fused position-wise feed-forward (csrc/ffn_f32.hip) ------------------------------------------
FUSED_FFN = os.environ.get("EAMD_FUSED_FFN", "1") != "0"
FUSED_FFN_MIN_ROWS = int(os.environ.get("EAMD_FUSED_FFN_MIN_ROWS", "4096"))     # 32 rows per workgroup: fewer rows leave CUs idle


def ffn_fused_ok(x, w1, w2, act):
    """True if eamd_ffn_fwd / _bwd take this problem (operands all fp32 in fp32 mode or all bf16 in bf16 mode, D = 256, F a
    multiple of 128 / 256, enough rows to fill the chip); otherwise the block runs as two eamd_gemm products"""
    bf = x.dtype == torch.bfloat16
    if not FUSED_FFN or x.dtype != w1.dtype or w1.dtype != w2.dtype or (bf and _state["precision"] != 1) or \
            (not bf and (x.dtype != torch.float32 or _state["precision"] != 0)):
        return False
    return (x.shape[1] == 256 and w1.shape[1] == 256 and w1.shape[0] % (256 if bf else 128) == 0 and w1.shape[0] >= 256
            and tuple(w2.shape) == (256, w1.shape[0]) and act in (ACT_RELU, ACT_SWISH)
            and (x.shape[0] >= FUSED_FFN_MIN_ROWS or ffn_hsplit(x.shape[0], w1.shape[0], bf) > 1))


# fp32 operands, few rows (the decoder's 3232 target positions = 101 row blocks on 256 CUs): two workgroups per row block, each over
# half of the hidden units, adding their halves of the second product into a zeroed result (eamd_ffn_t.hsplit).  Measured at
# config 2's decoder: one workgroup per block 147 us whatever the grid, the GEMM pair 104 us, split in two 77 us.
FUSED_FFN_HSPLIT_MIN_ROWS = int(os.environ.get("EAMD_FUSED_FFN_HSPLIT_MIN_ROWS", "1536"))


def ffn_hsplit(M, F, bf):
    if bf or M >= FUSED_FFN_MIN_ROWS or M < FUSED_FFN_HSPLIT_MIN_ROWS or F % 256 != 0 or F < 512:
        return 1
    return 2


def _ffn_desc(x, w1, b1, w2, b2, R, out, f, h, act, alpha, drop, F=None):
    p = _lib.FfnT()
    p.x, p.w1, p.b1, p.w2, p.b2, p.R = ptr(x), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(R)
    p.out, p.f, p.h = ptr(out), ptr(f), ptr(h)
    p.M, p.D, p.F, p.act = x.shape[0], x.shape[1], (F if F is not None else w1.shape[0]), act
    p.alpha = alpha
    p.dtype = 1 if x.dtype == torch.bfloat16 else 0
    p_in, s_in, p_out, s_out = drop
    p.p_in, p.salt_in, p.p_out, p.salt_out = float(p_in), int(s_in), float(p_out), int(s_out)
    if p_in > 0.0 or p_out > 0.0:
        p.drop_step = ptr(rng_state(x.device))
    return p


def _ffn_call(name, p, keep):
    fn = getattr(_lib.lib(), name)
    if _gemm_record is not None:
        _gemm_record.append((dict(kind=name, flop=4.0 * p.M * p.D * p.F), keep, lambda sp, p=p: check(fn(C.byref(p), sp), name)))
    check(fn(C.byref(p), stream_ptr()), name)


FFN_LN_FUSED = os.environ.get("EAMD_FFN_LN", "1") != "0"      # the LayerNorm in front of a fused FFN runs inside its kernel


def ffn_fwd(x, w1, b1, w2, b2, *, act, alpha=1.0, R=None, drop=(0.0, 0, 0.0, 0), save=True, packed=None, ln=None):
    """out = R + alpha * drop_out(drop_in(act(x W1^T + b1)) W2^T + b2) in ONE launch; with save also the two tensors
    backward needs: h = drop_in(act(z)) and f = mask / (1 - p) * act'(z).  -> (out fp32, f, h); x / w1 / w2 all fp32 or all
    bf16 (f, h in that dtype).
    ln = (x_raw fp32 [M, D], gamma, beta, eps, mean [M], rstd [M]): the block input is LayerNorm(x_raw), formed by the kernel
    while it stages its rows; `x` is then an uninitialised [M, D] buffer of the operand dtype that RECEIVES the normalised
    rows, and mean / rstd receive the row statistics (what eamd_layernorm_fwd would have left for backward)."""
    M, D = x.shape
    F = w1.shape[0]
    dt = x.dtype
    for t_ in (x, w1, w2):
        if t_.dtype != dt or dt not in (torch.float32, torch.bfloat16) or not t_.is_contiguous():
            raise _lib.EamdError("ffn_fwd: contiguous float32 or bfloat16 operands of one dtype")
    for t_ in tuple(v for v in (b1, b2, R) if v is not None):
        if t_.dtype != torch.float32 or not t_.is_contiguous():
            raise _lib.EamdError("ffn_fwd: biases / residual are contiguous float32 tensors")
    hs = ffn_hsplit(M, F, dt == torch.bfloat16)
    out = zeros(M, D, device=x.device) if hs > 1 else torch.empty(M, D, device=x.device, dtype=torch.float32)
    f = torch.empty(M, F, device=x.device, dtype=dt) if save else None
    h = torch.empty(M, F, device=x.device, dtype=dt) if save else None
    if packed is None:            # the kernels read the packed images (ffn_pack)
        packed = ffn_pack(w1, w2)[:2]
    p = _ffn_desc(x, packed[0], b1, packed[1], b2, R, out, f, h, act, alpha, drop, F=F)
    p.hsplit = hs
    if ln is not None:
        xr, g, b, eps, mean, rstd = ln
        for t_ in (xr, g, b, mean, rstd):
            if t_.dtype != torch.float32 or not t_.is_contiguous():
                raise _lib.EamdError("ffn_fwd: LayerNorm operands are contiguous float32 tensors")
        if tuple(xr.shape) != (M, D) or g.numel() != D or b.numel() != D or mean.numel() < M or rstd.numel() < M:
            raise _lib.EamdError("ffn_fwd: LayerNorm operand shapes")
        p.ln_x, p.ln_w, p.ln_b, p.ln_mean, p.ln_rstd, p.ln_eps = ptr(xr), ptr(g), ptr(b), ptr(mean), ptr(rstd), float(eps)
    _ffn_call("eamd_ffn_fwd", p, (x, packed, b1, b2, R, out, f, h, ln))
    return out, f, h


def ffn_pack(w1, w2):
    """the four packed weight images of one FFN (eamd_ffn_pack_f32 / _bf16) -> (fwd_first, fwd_second, bwd_first, bwd_second),
    each F * D elements in the operand dtype of the current mode.  Cached buffers on w1; the PACKING is redone at every call:
    the optimizer kernel rewrites the weights (and the bf16 shadow arena) in place - no tensor version to watch - and under
    hipGraph capture it has to be part of the replayed step."""
    F, D = w1.shape
    bf = fast() or w1.dtype == torch.bfloat16
    s1, s2 = (wshadow(w1), wshadow(w2)) if w1.dtype != torch.bfloat16 else (w1, w2)
    dt = torch.bfloat16 if bf else torch.float32
    buf = getattr(w1, "_eamd_ffn_pack", None)
    if buf is None or buf.numel() != 4 * F * D or buf.device != s1.device or buf.dtype != dt:
        buf = torch.empty(4, F * D, device=s1.device, dtype=dt)
        w1._eamd_ffn_pack = buf
    elif _rp["active"] and getattr(w1, "_eamd_ffn_pack_epoch", -1) == _rp["epoch"]:
        return buf[0], buf[1], buf[2], buf[3]         # packed by this pass's ffn_prepack (one launch for all blocks)
    name = "eamd_ffn_pack_bf16" if bf else "eamd_ffn_pack_f32"
    check(getattr(_lib.lib(), name)(ptr(s1), ptr(s2), ptr(buf[0]), ptr(buf[1]), ptr(buf[2]), ptr(buf[3]), D, F, stream_ptr()), name)
    return buf[0], buf[1], buf[2], buf[3]


def ffn_prepack(pairs):
    """pairs: list of (w1 [F, 256], w2 [256, F]) fp32 parameters: the packed images of ALL these feed-forward blocks in one
    launch (eamd_ffn_pack_f32_multi); ffn_pack() hands them out until rowproj_prepack_end().  fp32 mode only (no-op otherwise)."""
    if fast() or not pairs or os.environ.get("EAMD_FFN_F32_FORM", "") == "sym" or not _rp["active"]:
        return
    arr = (_lib.FfnPackT * len(pairs))()
    n = 0
    for w1, w2 in pairs:
        F, D = w1.shape
        if D != 256 or F % 128 != 0 or F < 256 or w1.dtype != torch.float32:
            continue
        buf = getattr(w1, "_eamd_ffn_pack", None)
        if buf is None or buf.numel() != 4 * F * D or buf.device != w1.device or buf.dtype != torch.float32:
            buf = torch.empty(4, F * D, device=w1.device, dtype=torch.float32)
            w1._eamd_ffn_pack = buf
        q = arr[n]
        q.w1, q.w2, q.fwd_first, q.fwd_second, q.bwd_first, q.bwd_second = ptr(w1), ptr(w2), ptr(buf[0]), ptr(buf[1]), ptr(buf[2]), ptr(buf[3])
        q.D, q.F = D, F
        w1._eamd_ffn_pack_epoch = _rp["epoch"]
        n += 1
    if n:
        check(_lib.lib().eamd_ffn_pack_f32_multi(arr, n, stream_ptr()), "eamd_ffn_pack_f32_multi")


def ffn_bwd_lnb_ok(M, F, dt):
    """eamd_ffn_bwd can run the LayerNorm backward as its epilogue (fp32 operands, one workgroup per row block)"""
    return dt == torch.float32 and ffn_hsplit(M, F, False) == 1 and os.environ.get("EAMD_FFN_F32_FORM", "") != "sym"


def ffn_bwd(dy, w1, w2, f, *, alpha=1.0, packed=None, lnb=None):
    """dz = alpha * (dy W2) (.) f and dx = dz W1 in ONE launch -> (dz [M, F], dx [M, D] fp32); packed = (bwd_first,
    bwd_second) of ffn_pack(w1, w2) (made here when not given).
    lnb = (x_block_input, gamma, mean, rstd, dres or None, ws, drop_out or None, (p, salt) or None): the LayerNorm backward runs
    on the dx rows inside the launch (as ops.rowproj's lnb): dx = LayerNorm'(dz W1) + dres."""
    M, D = dy.shape
    F = w1.shape[0]
    dt = dy.dtype
    assert f.shape == (M, F) and f.dtype == dt and f.is_contiguous() and dy.is_contiguous()
    dz = torch.empty(M, F, device=dy.device, dtype=dt)
    hs = ffn_hsplit(M, F, dt == torch.bfloat16)
    dx = zeros(M, D, device=dy.device) if hs > 1 else torch.empty(M, D, device=dy.device, dtype=torch.float32)
    if packed is None:
        packed = ffn_pack(w1, w2)[2:]
    assert packed[0].numel() == F * D and packed[1].numel() == F * D and packed[0].dtype == dt
    p = _ffn_desc(dy, packed[0], None, packed[1], None, None, dx, f, dz, ACT_NONE, alpha, (0.0, 0, 0.0, 0), F=F)
    p.hsplit = hs
    if lnb is not None:
        xb, g, mean, rstd, dres, ws, d_out, d_ps = lnb
        p.lnb_x, p.lnb_gamma, p.lnb_mean, p.lnb_rstd, p.lnb_dres, p.lnb_ws = ptr(xb), ptr(g), ptr(mean), ptr(rstd), ptr(dres), ptr(ws)
        if d_out is not None:
            p.lnb_drop_out, p.lnb_drop_p, p.lnb_drop_salt, p.drop_step = ptr(d_out), float(d_ps[0]), int(d_ps[1]), ptr(rng_state(dy.device))
    _ffn_call("eamd_ffn_bwd", p, (dy, packed, f, dz, dx, lnb))
    return dz, dx


# ---- row-block projections (csrc/rowproj_f32.hip) ---------------------------------------------------
ROWPROJ = os.environ.get("EAMD_ROWPROJ", "1") != "0"
ROWPROJ_MIN_ROWS = int(os.environ.get("EAMD_ROWPROJ_MIN_ROWS", "4096"))       # 32 rows per workgroup: fewer rows leave CUs idle


def rowproj_ok(M, K, N):
    """shapes eamd_rowproj takes in the current mode (fp32 operands; K, N multiples of 256, K <= 768; enough rows to fill the chip)"""
    return (ROWPROJ and _state["precision"] == 0 and M >= ROWPROJ_MIN_ROWS and K % 256 == 0 and N % 256 == 0 and 256 <= K <= 768
            and N <= 1024)


_rp = {"epoch": 0, "active": False}
_rp_bufs = {}


def rowproj_prepack(groups):
    """groups: list of (holder tensor, [(W, trans), ...]): the images of MANY blocks in one or two launches (an encoder packs all
    its layers' projection weights at the start of its forward); rowproj_images() then hands them out until rowproj_prepack_end()"""
    _rp["epoch"] += 1
    _rp["active"] = True
    flat = [job for _, jobs in groups for job in jobs]
    if not flat:
        return
    imgs = rowproj_pack(flat)
    i = 0
    for holder, jobs in groups:
        holder._eamd_rp = (_rp["epoch"], [id(w) for w, _ in jobs], imgs[i:i + len(jobs)])
        i += len(jobs)
    _rp["active"] = True


def rowproj_prepack_end():
    _rp["active"] = False


def rowproj_images(holder, jobs):
    """packed images of `jobs` = [(W, trans), ...] (the projection weights of one block, cached on `holder`): those of the
    running forward's prepack if there was one, else packed here (one launch)"""
    c = getattr(holder, "_eamd_rp", None)
    if _rp["active"] and c is not None and c[0] == _rp["epoch"] and len(c[2]) == len(jobs):
        return c[2]
    return rowproj_pack(jobs)


def rowproj_pack(jobs):
    """jobs: list of (W 2-D contiguous fp32 tensor (or view with a row stride), trans) -> list of packed images (K * N floats each),
    ALL made by one launch.  trans = False: W is [N, K] (y = x W^T); trans = True: W is [K, N] (dx = dy W).  The image buffers are
    cached on the weight tensor; the packing itself is redone at every call (the optimizer rewrites the weights in place, and
    under hipGraph capture it has to be part of the replayed step)."""
    arr = (_lib.RowProjPackT * len(jobs))()
    imgs = []
    for q, (W, trans) in zip(arr, jobs):
        assert W.dim() == 2 and W.dtype == torch.float32 and W.stride(1) == 1
        K, N = (W.shape[0], W.shape[1]) if trans else (W.shape[1], W.shape[0])
        # image buffers are cached per (weight storage address, shape, orientation): the weights live in persistent arenas
        key = (W.data_ptr(), K, N, int(bool(trans)), W.device.index)
        img = _rp_bufs.get(key)
        if img is None:
            if len(_rp_bufs) > 4096:
                _rp_bufs.clear()
            img = _rp_bufs[key] = torch.empty(K * N, device=W.device, dtype=torch.float32)
        q.w, q.image, q.K, q.N, q.ldw, q.trans = ptr(W), ptr(img), K, N, W.stride(0), int(bool(trans))
        imgs.append(img)
    check(_lib.lib().eamd_rowproj_pack_f32(arr, len(jobs), stream_ptr()), "eamd_rowproj_pack_f32")
    return imgs


def rowproj(a, img, N, *, bias=None, R=None, alpha=1.0, drop=None, ln=None, affine=None, lnb=None, out=None):
    """out[M, N] = R + alpha * dropout(A' B + bias), 32 rows per workgroup through the whole product (eamd_rowproj).
    a [M, K] fp32 (row stride allowed); img = rowproj_pack image of B [K, N].
    ln = (x_raw [M, 256], gamma, beta, eps, mean [M], rstd [M]): A' = LayerNorm(x_raw), `a` RECEIVES the normalised rows.
    affine = (scale [256], shift [256], act, a_out or None): A' = act(a * scale + shift), also written to a_out.
    lnb = (x_block_input [M, 256], gamma, mean, rstd, dres or None, ws, drop_out or None, (p, salt) or None): N = 256, the
    LayerNorm backward runs on the result rows: out = dx (+ dres), ws = per-workgroup partials of d gamma / d beta."""
    M, K = a.shape
    dev = a.device
    if out is None:
        out = torch.empty(M, N, device=dev, dtype=torch.float32)
    p = _lib.RowProjT()
    p.a, p.lda, p.w, p.bias = ptr(a), a.stride(0), ptr(img), ptr(bias)
    p.R, p.ldr = ptr(R), (R.stride(0) if R is not None else 0)
    p.out, p.ldo = ptr(out), out.stride(0)
    p.M, p.K, p.N, p.alpha = M, K, N, float(alpha)
    if drop is not None and drop[0] > 0.0:
        p.p_out, p.salt_out, p.drop_step = float(drop[0]), int(drop[1]), ptr(rng_state(dev))
    if ln is not None:
        xr, g, b, eps, mean, rstd = ln
        p.ln_x, p.ln_w, p.ln_b, p.ln_mean, p.ln_rstd, p.ln_eps = ptr(xr), ptr(g), ptr(b), ptr(mean), ptr(rstd), float(eps)
    if affine is not None:
        sc, sh, act, a_out = affine
        p.a_scale, p.a_shift, p.a_act, p.a_out = ptr(sc), ptr(sh), int(act), ptr(a_out)
    if lnb is not None:
        xb, g, mean, rstd, dres, ws, d_out, d_ps = lnb
        p.lnb_x, p.lnb_gamma, p.lnb_mean, p.lnb_rstd, p.lnb_dres, p.lnb_ws = ptr(xb), ptr(g), ptr(mean), ptr(rstd), ptr(dres), ptr(ws)
        if d_out is not None:
            p.lnb_drop_out, p.lnb_drop_p, p.lnb_drop_salt, p.drop_step = ptr(d_out), float(d_ps[0]), int(d_ps[1]), ptr(rng_state(dev))
    fn = _lib.lib().eamd_rowproj
    keep = (a, img, bias, R, out, ln, affine, lnb)
    if _gemm_record is not None:
        _gemm_record.append((dict(kind="rowproj", flop=2.0 * M * K * N), keep, lambda sp, p=p: check(fn(C.byref(p), sp), "eamd_rowproj")))
    check(fn(C.byref(p), stream_ptr()), "eamd_rowproj")
    return out


def rowproj_lnb_ws(M, device):
    return torch.empty(int(_lib.lib().eamd_rowproj_lnb_workspace(M)), device=device, dtype=torch.float32)


# ---- weight-gradient side stream ---------------------------------------------------------------
# dW = dY^T X is needed only by the optimizer (and the gradient all-reduce), never by the rest of
# backward.  When enabled, every weight-gradient GEMM is issued on a second HIP stream so that it
# overlaps the dX chain; under hipGraph capture this becomes a parallel branch of the graph.
_wgrad = {"stream": None, "used": False}


def enable_wgrad_stream(enable=True):
    _wgrad["stream"] = torch.cuda.Stream() if enable else None
    _wgrad["used"] = False


def wgrad_group_end():
    wgrad_group_flush()
    _wgroup["on"] = False


def wgrad_join():
    """launch the queued weight-gradient GEMMs and make the current stream wait for all weight-gradient work issued so far"""
    wgrad_group_flush()
    if _wgroup.get("side_used"):
        torch.cuda.current_stream().wait_stream(_wgroup["side"])
        _wgroup["side_used"] = False
        _wgroup["held"] = []
    st = _wgrad["stream"]
    if st is not None and _wgrad["used"]:
        torch.cuda.current_stream().wait_stream(st)
        _wgrad["used"] = False


def linear_bwd_w(dy, x, dW, *, alpha=1.0, b_act=ACT_NONE, db=None, a_drop=None, b_drop=None, group=True):
    """group=False: never queued for the grouped launch (per-chunk gradients of a streamed loss: queueing would keep
    every chunk's operands alive until the flush)"""
    st = _wgrad["stream"]
    if st is None:
        return _linear_bwd_w(dy, x, dW, alpha=alpha, b_act=b_act, db=db, a_drop=a_drop, b_drop=b_drop, group=group)
    cur = torch.cuda.current_stream()
    st.wait_stream(cur)                 # operands were produced on the main stream
    with torch.cuda.stream(st):
        _linear_bwd_w(dy, x, dW, alpha=alpha, b_act=b_act, db=db, a_drop=a_drop, b_drop=b_drop)
    dy.record_stream(st)                # keep the caching allocator from recycling them early
    x.record_stream(st)
    _wgrad["used"] = True


def _linear_bwd_w(dy, x, dW, *, alpha=1.0, b_act=ACT_NONE, db=None, a_drop=None, b_drop=None, group=True):
    """dW[N,K] += alpha * drop_a(dy)[M,N]^T @ drop_b(b_act(x))[M,K]   (split-K, f32 atomics)
    db[N] += alpha * column sums of drop_a(dy) (bias gradient, fused into the same launch);
    a_drop / b_drop = (p, salt): fp32 mode only, the operand is dropped while it is staged (eamd_gemm_t.a_drop_p)"""
    M, N = dy.shape
    K = x.shape[1]
    assert x.shape[0] == M and dW.numel() == N * K and dW.dtype == torch.float32
    grouping = group and _wgroup["on"] and _wgrad["stream"] is None
    if (grouping and M <= STACK_WGRAD_MAX_ROWS and alpha == 1.0 and b_act == ACT_NONE
            and a_drop is None and b_drop is None and dy.is_contiguous() and x.is_contiguous() and _in_grad_arena(dW)):
        # one decoder step's weight gradient (M = batch rows) reads and writes the whole dW for a reduction of M: the
        # steps of a backward pass are stacked along the reduction instead and leave as ONE product at the flush
        # (config 4: 101 x (22 + 12 us) of 32 MB / 16 MB read-modify-writes -> two GEMMs with K = 3232)
        key = (dW.data_ptr(), db.data_ptr() if db is not None else 0, dy.dtype, x.dtype, N, K)
        _wgroup["stack"].setdefault(key, (dW, db, [], []))
        ent = _wgroup["stack"][key]
        ent[2].append(dy)
        ent[3].append(x)
        return
    sk = auto_splitk(N, K, M)
    tile = 0
    t64 = ((N + 63) // 64) * ((K + 63) // 64)
    if grouping and t64 <= GROUP_WGRAD_MAX_TILES and _in_grad_arena(dW):
        if t64 >= GROUP_WGRAD_TILE128_MIN and N >= 128 and K >= 128 and dy.dtype == torch.float32 and _state["precision"] == 0:
            # big outputs (the FFN weights) go to the 128x128 grouped launch: a quarter of the operand re-reads
            tile = 128
            sk = max(1, min(GROUP_WGRAD_T128_WGS // (((N + 127) // 128) * ((K + 127) // 128)), M // 256))
        else:
            tile = 64
            sk = max(1, sk // GROUP_WGRAD_SK_DIV)
    gemm(dy, x, dW, N, K, M, N, K, K, transA=1, transB=1, alpha=alpha, b_act=b_act, splitk=sk, tile=tile,
         beta=1.0 if sk == 1 else 0.0, colsum=db, a_drop=a_drop, b_drop=b_drop, group=grouping)


def colsum(x, out, scale=1.0, rows=None, D=None, ld=None):
    rows = x.shape[0] if rows is None else rows
    D = x.shape[1] if D is None else D
    ld = D if ld is None else ld
    assert out.numel() >= D
    check(_lib.lib().eamd_colsum(ptr(x), C.c_int64(ld), ptr(out), C.c_int64(rows), D, C.c_float(scale),
                                 1 if x.dtype == torch.bfloat16 else 0, stream_ptr()), "eamd_colsum")


# ---- row kernels -------------------------------------------------------------------------------
def layernorm_fwd(x, gamma, beta, eps, out_dtype=torch.float32):
    rows, D = x.shape
    y = torch.empty(rows, D, device=x.device, dtype=out_dtype)
    mean = torch.empty(rows, device=x.device, dtype=torch.float32)
    rstd = torch.empty(rows, device=x.device, dtype=torch.float32)
    y32, y16 = (ptr(y), None) if out_dtype == torch.float32 else (None, ptr(y))
    check(_lib.lib().eamd_layernorm_fwd(ptr(x), ptr(gamma), ptr(beta), y32, y16, ptr(mean), ptr(rstd), rows, D,
                                        C.c_float(eps), stream_ptr()), "eamd_layernorm_fwd")
    return y, mean, rstd


class _LnReduceJob(C.Structure):
    _fields_ = [("ws", C.c_void_p), ("dgamma", C.c_void_p), ("dbeta", C.c_void_p), ("nblk", C.c_int32), ("D", C.c_int32)]


# Deferred gamma / beta reduction: inside a backward pass every LayerNorm backward leaves its per-block partial sums
# in its workspace and ONE launch at the end of the pass (autograd's final callback, the hook DistributedDataParallel
# uses) adds them all into the gradients - 80 launches per config-2 training step become one.  Only for gradients
# that live in the flat arena of train.FlatParams (persistent memory); switched off while a
# GradReducer overlaps bucket all-reduces with backward (the gradients must be final when a block reports them).
defer_ln_reduce = True
_ln_pending = []
_ln_task = -1        # autograd graph-task id the pending partials belong to


def flush_ln_reduce():
    """add the pending LayerNorm gamma / beta partials into their gradients (one launch per 64 LayerNorms)"""
    global _ln_pending, _ln_task
    jobs, _ln_pending, _ln_task = _ln_pending, [], -1
    if not jobs:
        return
    tab = (_LnReduceJob * len(jobs))()
    for i, (ws, dg, db, nblk, D) in enumerate(jobs):
        tab[i].ws, tab[i].dgamma, tab[i].dbeta, tab[i].nblk, tab[i].D = ws.data_ptr(), dg.data_ptr(), db.data_ptr(), nblk, D
    with torch.cuda.device(jobs[0][0].device):
        check(_lib.lib().eamd_layernorm_bwd_reduce(tab, len(jobs), stream_ptr()), "eamd_layernorm_bwd_reduce")


def ln_partials_reduce(ws, dgamma, dbeta, nblk, D):
    """per-block partial sums ws [nblk][2][D] of a LayerNorm backward (left by eamd_rowproj's epilogue) -> dgamma / dbeta: queued
    behind the running backward pass like every other LayerNorm's second stage, or reduced right away outside one"""
    if _defer_ln(ws, dgamma, dbeta, nblk, D):
        return
    tab = (_LnReduceJob * 1)()
    tab[0].ws, tab[0].dgamma, tab[0].dbeta, tab[0].nblk, tab[0].D = ws.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), nblk, D
    check(_lib.lib().eamd_layernorm_bwd_reduce(tab, 1, stream_ptr()), "eamd_layernorm_bwd_reduce")


def _defer_ln(ws, dgamma, dbeta, nblk, D):
    """True if the reduction of this pass was queued behind the running backward pass"""
    global _ln_pending, _ln_task
    if not defer_ln_reduce or not (getattr(dgamma, "_eamd_arena", False) and getattr(dbeta, "_eamd_arena", False)):
        return False      # gradients handed back to autograd by value must be complete when backward returns them
    task = torch._C._current_graph_task_id()
    if task < 0:                  # not inside a backward pass (direct call): reduce right away
        return False
    if task != _ln_task:          # first LayerNorm of this pass (partials of a pass that died are dropped)
        _ln_pending = []
        torch.autograd.Variable._execution_engine.queue_callback(flush_ln_reduce)
        _ln_task = task
    _ln_pending.append((ws, dgamma, dbeta, nblk, D))
    return True


def layernorm_bwd(dy, x, gamma, mean, rstd, dres, dgamma, dbeta, drop=None):
    """drop = (p, salt): also return bf16(dropout(dx; p, salt)) written by the same kernel (D = 256 / 512 only) ->
    (dx, dx_dropped); otherwise -> dx"""
    rows, D = x.shape
    assert dy.shape == x.shape and dgamma.numel() == D and dbeta.numel() == D
    dx = torch.empty_like(x)
    L = _lib.lib()
    nws = int(L.eamd_layernorm_bwd_workspace(rows, D))
    ws = torch.empty(nws, device=x.device, dtype=torch.float32)
    pg, pb = ptr(dgamma), ptr(dbeta)
    if _defer_ln(ws, dgamma, dbeta, nws // (2 * D), D):
        pg = pb = None
    if drop is not None:
        dx16 = torch.empty(rows, D, device=x.device, dtype=act_dtype())        # dropped copy in the GEMM-operand dtype
        fn, name = ((L.eamd_layernorm_bwd_drop, "eamd_layernorm_bwd_drop") if fast()
                    else (L.eamd_layernorm_bwd_drop_f32, "eamd_layernorm_bwd_drop_f32"))
        check(fn(ptr(dy), ptr(x), ptr(gamma), ptr(mean), ptr(rstd), ptr(dres), ptr(dx), ptr(dx16),
                 C.c_float(drop[0]), ptr(rng_state(x.device)), C.c_uint64(drop[1]), pg,
                 pb, ptr(ws), rows, D, stream_ptr()), name)
        return dx, dx16
    check(L.eamd_layernorm_bwd(ptr(dy), ptr(x), ptr(gamma), ptr(mean), ptr(rstd), ptr(dres), ptr(dx),
                               pg, pb, ptr(ws), rows, D, stream_ptr()), "eamd_layernorm_bwd")
    return dx


F32_FUSED_ATTN = True   # fp32 mode: eamd_attn_fwd_f32 / eamd_attn_bwd_q_f32 (tests flip it to reach the GEMM / softmax path)


def attn_fwd_supported(T1, T2, dk, rel):
    """shapes the fused attention forward (eamd_attn_fwd) covers; alignment is checked by the library"""
    if dk != 64 or (rel and T1 != T2):
        return False
    # rows of 513 .. 4096 keys: the key-split long-row kernels (attn_f32.hip; 16 queries per workgroup up to 2048 keys, 8 beyond)
    return (fast() or F32_FUSED_ATTN) and T2 <= 4096


def attn_fwd(qu, qv, k, v, pos, mask, B, T1, T2, H, dk, ldp, scale, drop=None, shift_len=None):
    """qu / qv / k / v: (tensor, element offset, row stride) views of [rows, *] matrices, heads side by side, all bf16
    (eamd_attn_fwd) or all fp32 (eamd_attn_fwd_f32); pos: such a view of the [T2, H*dk] projected positions, or None.
    Returns (P [H*B*T1*ldp], Pd, ctx [B*T1, H*dk]) in that dtype, or None if the library declines the operands
    (EAMD_EUNSUPPORTED).  drop = (p, salt): attention dropout - ctx is built from Pd = dropout(P) (same mask as
    ops.dropout(P, p, salt)), which is returned beside the undropped P; without dropout Pd is P."""
    dev = qu[0].device
    dt = qu[0].dtype
    D = H * dk
    mb = mi = 0
    if mask is not None:
        assert mask.dtype == torch.uint8 and mask.is_contiguous() and mask.dim() == 3 and mask.shape[0] in (1, B)
        assert mask.shape[2] == T2 and mask.shape[1] in (1, T1)
        mb = 0 if mask.shape[0] == 1 else mask.shape[1] * mask.shape[2]
        mi = 0 if mask.shape[1] == 1 else T2
    assert dt in (torch.bfloat16, torch.float32)
    for t_, _, _ in (qu, k, v) + ((qv,) if qv is not None else ()):
        assert t_.dtype == dt and t_.is_cuda
    assert pos is None or (pos[0].dtype == dt and pos[2] >= D)
    P = torch.empty(H * B * T1 * ldp, device=dev, dtype=dt)
    cx = torch.empty(B * T1, D, device=dev, dtype=dt)
    dropping = drop is not None and drop[0] > 0.0
    Pd = torch.empty_like(P) if dropping else P
    dargs = ((ptr(Pd), C.c_float(drop[0]), ptr(rng_state(dev)), C.c_uint64(drop[1])) if dropping
             else (None, C.c_float(0.0), None, C.c_uint64(0)))
    i64 = C.c_int64
    args = (ptr(qu[0], qu[1]), i64(qu[2]), ptr(qv[0], qv[1]) if qv is not None else None, i64(qv[2] if qv is not None else 0),
            ptr(k[0], k[1]), i64(k[2]), ptr(v[0], v[1]), i64(v[2]), ptr(pos[0], pos[1]) if pos is not None else None,
            i64(pos[2] if pos is not None else 0),
            ptr(mask), i64(mb), i64(mi), ptr(P), i64(ldp), ptr(cx), i64(D), B, H, T1, T2, dk, C.c_float(scale)) + dargs
    # relative positions on a padded batch: the rel_shift is taken over the batch's own length (shift_len: int32 device scalar)
    args = args + (ptr(shift_len) if (pos is not None and T1 == T2) else None,)
    name = "eamd_attn_fwd" if dt == torch.bfloat16 else "eamd_attn_fwd_f32"
    fn = getattr(_lib.lib(), name)
    rc = fn(*args, stream_ptr())
    if rc == _lib.EAMD_EUNSUPPORTED:
        return None
    check(rc, name)
    if _gemm_record is not None:
        _gemm_record.append((dict(kind=name, flop=0.0), (qu, qv, k, v, pos, mask, P, Pd, cx), lambda sp, args=args: check(fn(*args, sp), name)))
    return P, Pd, cx


def attn_bwd_q(dctx, k, v, P, dS, dbd, dq, B, T1, T2, H, dk, ldp, scale, drop=None, shift_len=None):
    """dctx / k / v / dq: (tensor, element offset, row stride) views; P, dS (and dbd or None): [H*B*T1*ldp], bf16
    (eamd_attn_bwd_q; dq fp32 or bf16) or everything fp32 (eamd_attn_bwd_q_f32).
    dS, dbd and dq are written.  Returns False if the library declines the operands (EAMD_EUNSUPPORTED)."""
    i64 = C.c_int64
    dropping = drop is not None and drop[0] > 0.0      # the forward's attention dropout: dP <- mask * dP / (1 - p)
    dargs = ((C.c_float(drop[0]), ptr(rng_state(P.device)), C.c_uint64(drop[1])) if dropping
             else (C.c_float(0.0), None, C.c_uint64(0)))
    if P.dtype == torch.bfloat16:
        name = "eamd_attn_bwd_q"
        args = (ptr(dctx[0], dctx[1]), i64(dctx[2]), ptr(k[0], k[1]), i64(k[2]), ptr(v[0], v[1]), i64(v[2]), ptr(P), i64(ldp),
                ptr(dS), ptr(dbd), ptr(dq[0], dq[1]), i64(dq[2]), int(dq[0].dtype == torch.bfloat16), B, H, T1, T2, dk,
                C.c_float(scale)) + dargs
    else:
        name = "eamd_attn_bwd_q_f32"
        for t_ in (dctx[0], k[0], v[0], P, dS, dq[0]) + ((dbd,) if dbd is not None else ()):
            assert t_.dtype == torch.float32
        args = (ptr(dctx[0], dctx[1]), i64(dctx[2]), ptr(k[0], k[1]), i64(k[2]), ptr(v[0], v[1]), i64(v[2]), ptr(P), i64(ldp),
                ptr(dS), ptr(dbd), ptr(dq[0], dq[1]), i64(dq[2]), B, H, T1, T2, dk, C.c_float(scale)) + dargs
    args = args + (ptr(shift_len) if (dbd is not None and T1 == T2) else None,)
    fn = getattr(_lib.lib(), name)
    rc = fn(*args, stream_ptr())
    if rc == _lib.EAMD_EUNSUPPORTED:
        return False
    check(rc, name)
    if _gemm_record is not None:
        _gemm_record.append((dict(kind=name, flop=0.0), (dctx, k, v, P, dS, dbd, dq), lambda sp, args=args: check(fn(*args, sp), name)))
    return True


FUSED_ATTN_KV = True   # fp32 mode: eamd_attn_bwd_kv_f32 (tests flip it to reach the batched GEMMs)


def attn_bwd_kv(Pd, dS, dbd, dctx, qu, qv, dv, dk_, dpos, B, T1, T2, H, dk, ldp):
    """dv = Pd^T dctx, dk = dS^T qu (and, fp32 only, dpos += dbd^T qv) in one launch.  All tensors fp32
    (eamd_attn_bwd_kv_f32) or all bf16 (eamd_attn_bwd_kv).  dctx / qu / qv / dv / dk_ / dpos: (tensor, element offset,
    row stride) views; dv and dk_ share their row stride.  Returns False if the library declines the operands
    (EAMD_EUNSUPPORTED)."""
    i64 = C.c_int64
    dt = Pd.dtype
    assert dt in (torch.float32, torch.bfloat16)
    for t_ in (Pd, dS, dctx[0], qu[0], dv[0], dk_[0]):
        assert t_.dtype == dt and t_.is_cuda
    assert dv[2] == dk_[2]
    rel = dbd is not None
    if dt == torch.float32:
        args = (ptr(Pd), ptr(dS), ptr(dbd) if rel else None, i64(ldp), ptr(dctx[0], dctx[1]), i64(dctx[2]), ptr(qu[0], qu[1]),
                i64(qu[2]), ptr(qv[0], qv[1]) if rel else None, i64(qv[2] if rel else 0), ptr(dv[0], dv[1]), ptr(dk_[0], dk_[1]),
                i64(dv[2]), ptr(dpos[0], dpos[1]) if rel else None, i64(dpos[2] if rel else 0), B, H, T1, T2, dk)
        name = "eamd_attn_bwd_kv_f32"
    else:
        assert not rel, "the bf16 key-side launch leaves the positional product to the GEMM"
        args = (ptr(Pd), ptr(dS), i64(ldp), ptr(dctx[0], dctx[1]), i64(dctx[2]), ptr(qu[0], qu[1]), i64(qu[2]),
                ptr(dv[0], dv[1]), ptr(dk_[0], dk_[1]), i64(dv[2]), B, H, T1, T2, dk)
        name = "eamd_attn_bwd_kv"
    fn = getattr(_lib.lib(), name)
    rc = fn(*args, stream_ptr())
    if rc == _lib.EAMD_EUNSUPPORTED:
        return False
    check(rc, name)
    if _gemm_record is not None:
        _gemm_record.append((dict(kind=name, flop=0.0), (Pd, dS, dbd, dctx, qu, qv, dv, dk_, dpos),
                             lambda sp, args=args: check(fn(*args, sp), name)))
    return True


def softmax_fwd(ac, bd, mask, P, nblocks, B, T1, T2, ld, scale):
    """P may be fp32 (may alias ac) or bf16 (separate buffer)"""
    mb = mi = 0
    if mask is not None:
        assert mask.dtype == torch.uint8 and mask.is_contiguous() and mask.dim() == 3 and mask.shape[0] in (1, B)
        assert mask.shape[2] == T2 and mask.shape[1] in (1, T1)
        mb = 0 if mask.shape[0] == 1 else mask.shape[1] * mask.shape[2]   # batch-broadcast masks
        mi = 0 if mask.shape[1] == 1 else T2
    assert ac.numel() >= nblocks * T1 * ld and P.numel() >= nblocks * T1 * ld
    p32, p16 = (ptr(P), None) if P.dtype == torch.float32 else (None, ptr(P))
    check(_lib.lib().eamd_softmax_fwd(ptr(ac), ptr(bd), ptr(mask), C.c_int64(mb), C.c_int64(mi), p32, p16, nblocks,
                                      B, T1, T2, C.c_int64(ld), C.c_float(scale), stream_ptr()),
          "eamd_softmax_fwd")


def softmax_bwd(P, dP, dbd, nblocks, T1, T2, ld, scale, dS16=None):
    """fp32: dP is overwritten with dS, dbd (fp32) receives the inverse rel-shift scatter.
    bf16: P bf16, dP fp32 in, dS16 (bf16) out, dbd bf16.  dbd is fully written by the kernel (no pre-zeroing)."""
    assert P.numel() >= nblocks * T1 * ld and dP.numel() >= nblocks * T1 * ld and dP.dtype == torch.float32
    if P.dtype == torch.bfloat16:
        assert dS16 is not None and dS16.dtype == torch.bfloat16 and (dbd is None or dbd.dtype == torch.bfloat16)
        check(_lib.lib().eamd_softmax_bwd(None, ptr(P), ptr(dP), None, ptr(dS16), ptr(dbd), nblocks, T1, T2,
                                          C.c_int64(ld), C.c_float(scale), stream_ptr()), "eamd_softmax_bwd")
    else:
        check(_lib.lib().eamd_softmax_bwd(ptr(P), None, ptr(dP), ptr(dbd), None, None, nblocks, T1, T2,
                                          C.c_int64(ld), C.c_float(scale), stream_ptr()), "eamd_softmax_bwd")


def lsm_loss(logits, target, smoothing, inv_denom, ignore_id, want_grad=True):
    rows, V = logits.shape
    assert target.numel() == rows and target.dtype == torch.int64
    loss_rows = torch.empty(rows, device=logits.device, dtype=torch.float32)
    correct = torch.empty(rows, device=logits.device, dtype=torch.float32)
    grad = torch.empty_like(logits) if want_grad else None
    check(_lib.lib().eamd_lsm_loss(ptr(logits), ptr(target), ptr(loss_rows), ptr(correct), ptr(grad), rows, V,
                                   ignore_id, C.c_float(smoothing), C.c_float(inv_denom), stream_ptr()),
          "eamd_lsm_loss")
    return loss_rows, correct, grad


def argmax_rows(x):
    rows, V = x.shape
    out = torch.empty(rows, device=x.device, dtype=torch.int32)
    check(_lib.lib().eamd_argmax_rows(ptr(x), C.c_int64(V), ptr(out), rows, V, stream_ptr()), "eamd_argmax_rows")
    return out


def reduce_sum(x, scale=1.0):
    out = torch.empty((), device=x.device, dtype=torch.float32)
    check(_lib.lib().eamd_reduce_sum(ptr(x), C.c_int64(x.numel()), ptr(out), C.c_float(scale), stream_ptr()),
          "eamd_reduce_sum")
    return out


def linear_rows_ln(x, gamma, beta, eps, W, b, *, act=EPI_NONE, R=None, alpha=1.0, out=None):
    """y = alpha * act(LayerNorm(x) W^T + b) + R in ONE launch (eamd_linear_rows_ln_f32: at most 16 rows, or up to 1024 rows at
    K <= 256 in 16-row blocks on the matrix cores), or None when the library declines (more rows, wider K, unaligned): the caller
    then normalises and multiplies in two launches"""
    M, K = x.shape
    N = W.shape[0]
    # (16-row blocks: every 16 x 16 tile's wave repeats the LayerNorm of its rows - measured at M = 320: 10.6 us against
    # LayerNorm + product = 4.7 + 7.8 us at N = 256 / 768, but 18.8 against 18.4 at N = 2048 and 40 against 22.5 at N = 5000)
    if (M > 16 and (K > 256 or M > LINEAR_ROWS_BLOCK_MAX or N > 1024)) or K > 1024 or x.dtype != torch.float32 or W.dtype != torch.float32 or x.stride(1) != 1 or act not in (EPI_NONE, EPI_RELU, EPI_SWISH):
        return None
    y = out if out is not None else torch.empty(M, N, device=x.device, dtype=torch.float32)
    rc = _lib.lib().eamd_linear_rows_ln_f32(ptr(x), ptr(gamma), ptr(beta), C.c_float(eps), ptr(W), ptr(b), ptr(R), ptr(y), M, N, K,
                                            int(act), C.c_float(alpha), C.c_int64(x.stride(0) if M > 1 else K),
                                            C.c_int64(R.stride(0) if (R is not None and M > 1) else 0), C.c_int64(y.stride(0)), stream_ptr())
    if rc == _lib.EAMD_EUNSUPPORTED:
        return None
    check(rc, "eamd_linear_rows_ln_f32")
    return y


def decode_self_attn(qkv, Kc, Vc, slot_at, pos, H, pos_dev=None):
    """the newest position of n hypotheses attends over its prefix; keys / values of this step (columns D.., 2D.. of qkv) are
    appended to the time-major caches Kc / Vc [Lcap, n, D] at row `pos`; slot_at [n, Lcap] int32 (eamd_decode_self_attn) -> ctx [n, D]"""
    n = qkv.shape[0]
    Lcap, n2, D = Kc.shape
    assert n2 == n and Vc.shape == Kc.shape and slot_at.shape == (n, Lcap) and slot_at.dtype == torch.int32 and qkv.stride(1) == 1
    ctx = torch.empty(n, D, device=qkv.device, dtype=torch.float32)
    check(_lib.lib().eamd_decode_self_attn_dyn(ptr(qkv), C.c_int64(qkv.stride(0)), ptr(Kc), ptr(Vc), ptr(slot_at), Lcap, int(pos),
                                               ptr(pos_dev), n, H, D, ptr(ctx), stream_ptr()), "eamd_decode_self_attn")
    return ctx


SRC_ATTN_SPLITS = int(os.environ.get("EAMD_SRC_ATTN_SPLITS", "4"))      # key splits of the grouped source attention (1: none)


def decode_src_attn(q, kv, k_off, v_off, ldkv, mask, G, g, T, H, group=False):
    """one query position per hypothesis over the memory of its utterance (eamd_decode_src_attn): q [G * g, D]; kv = the tensor
    that holds keys / values of this layer at element offsets k_off / v_off, row stride ldkv ([G * T] rows); mask [G, 1, T] uint8
    or None -> ctx [G * g, D], or None when the library declines.  group: one workgroup per (utterance, head) for all g
    hypotheses (eamd_decode_src_attn_group; g <= 16, T <= 1024) instead of one per (hypothesis, head)"""
    n, D = q.shape
    ctx = torch.empty(n, D, device=q.device, dtype=torch.float32)
    splits = SRC_ATTN_SPLITS if (group and T >= 128 and G * H * SRC_ATTN_SPLITS <= 4096) else 1
    if group and splits > 1:       # the keys of an utterance over several workgroups + a merge launch (eamd_decode_src_attn_split)
        L = _lib.lib()
        L.eamd_decode_src_attn_split_workspace.restype = C.c_int64
        ws = torch.empty(int(L.eamd_decode_src_attn_split_workspace(G, g, H, splits)), device=q.device, dtype=torch.float32)
        rc = L.eamd_decode_src_attn_split(ptr(q), C.c_int64(q.stride(0)), ptr(kv, k_off), ptr(kv, v_off), C.c_int64(ldkv), ptr(mask),
                                          G, g, T, H, D, splits, ptr(ws), ptr(ctx), stream_ptr())
        if rc != _lib.EAMD_EUNSUPPORTED:
            check(rc, "eamd_decode_src_attn_split")
            return ctx
    fn = _lib.lib().eamd_decode_src_attn_group if group else _lib.lib().eamd_decode_src_attn
    rc = fn(ptr(q), C.c_int64(q.stride(0)), ptr(kv, k_off), ptr(kv, v_off), C.c_int64(ldkv), ptr(mask),
                                         G, g, T, H, D, ptr(ctx), stream_ptr())
    if rc == _lib.EAMD_EUNSUPPORTED:
        return None
    check(rc, "eamd_decode_src_attn")
    return ctx


def beam_slots(slot_in, hyp, pos, pos_dev=None):
    """the slot table behind a beam step's selection: row i = row hyp[i] of slot_in with column `pos` set to hyp[i]
    (pos_dev: int32 device scalar added to pos)"""
    n, Lcap = slot_in.shape
    out = torch.empty_like(slot_in)
    check(_lib.lib().eamd_beam_slots_dyn(ptr(slot_in), ptr(out), ptr(hyp), n, Lcap, int(pos), ptr(pos_dev), stream_ptr()), "eamd_beam_slots")
    return out


def copy_jobs(pairs):
    """up to 16 small device-to-device copies (dst <- src, contiguous tensors of equal byte size) in one launch (eamd_copy_jobs)"""
    assert 0 < len(pairs) <= 16
    for d, s_ in pairs:
        assert d.is_contiguous() and s_.is_contiguous() and d.numel() * d.element_size() == s_.numel() * s_.element_size(), (d.shape, s_.shape)
    n = len(pairs)
    src = (C.c_void_p * n)(*[s_.data_ptr() for _, s_ in pairs])
    dst = (C.c_void_p * n)(*[d.data_ptr() for d, _ in pairs])
    nb = (C.c_int64 * n)(*[d.numel() * d.element_size() for d, _ in pairs])
    check(_lib.lib().eamd_copy_jobs(src, dst, nb, n, stream_ptr()), "eamd_copy_jobs")


def weighted_sum(logps, weights):
    """sum_k w_k logp_k over up to four contiguous fp32 [n, V] matrices, in torch's order ((0 + w_0 l_0) + w_1 l_1 ...): one launch"""
    out = torch.empty_like(logps[0])
    arr = (C.c_void_p * 4)(*[lp.data_ptr() for lp in logps] + [None] * (4 - len(logps)))
    wts = (C.c_float * 4)(*[float(w) for w in weights] + [0.0] * (4 - len(weights)))
    check(_lib.lib().eamd_weighted_sum(arr, wts, len(logps), C.c_int64(out.numel()), ptr(out), stream_ptr()), "eamd_weighted_sum")
    return out


def weighted_topk_rows(logps, weights, k, extra=-1):
    """weighted_sum + topk_rows(idx32=True) in one launch (eamd_weighted_topk_rows) -> (pre [n, V], idx int64 [n, k], idx32).
    extra >= 0: a column k with that token (-1 where it is already among the k) is appended to idx / idx32"""
    n, V = logps[0].shape
    ko = k + (1 if extra >= 0 else 0)
    for lp in logps:
        if lp.dtype != torch.float32 or not lp.is_contiguous() or lp.shape != (n, V):
            raise _lib.EamdError("weighted_topk_rows: contiguous float32 [n, V] matrices")
    pre = torch.empty_like(logps[0])
    vals = torch.empty(n, ko, device=pre.device, dtype=torch.float32)
    idx = torch.empty(n, ko, device=pre.device, dtype=torch.int64)
    i32 = torch.empty(n, ko, device=pre.device, dtype=torch.int32)
    arr = (C.c_void_p * 4)(*[lp.data_ptr() for lp in logps] + [None] * (4 - len(logps)))
    wts = (C.c_float * 4)(*[float(w) for w in weights] + [0.0] * (4 - len(weights)))
    check(_lib.lib().eamd_weighted_topk_rows(arr, wts, len(logps), n, V, k, int(extra), ptr(pre), ptr(vals), ptr(idx), ptr(i32), stream_ptr()),
          "eamd_weighted_topk_rows")
    return pre, idx, i32


def beam_select(pre, ids, psi, c_s, hyp, w_ctc, nutt, beam):
    """the `beam` best continuations of each utterance among its beam x P pre-beam candidates (eamd_beam_select) ->
    (top_s [nutt, beam], top_i [nutt, beam] = local slot * V + token, c_local [n, P] = psi - c_s)"""
    n, V = pre.shape
    P = ids.shape[1]
    assert n == nutt * beam and ids.dtype == torch.int64 and ids.is_contiguous() and psi.shape == (n, P) and pre.is_contiguous()
    top_s = torch.empty(nutt, beam, device=pre.device, dtype=torch.float32)
    top_i = torch.empty(nutt, beam, device=pre.device, dtype=torch.int64)
    c_local = torch.empty(n, P, device=pre.device, dtype=torch.float32)
    check(_lib.lib().eamd_beam_select(ptr(pre), ptr(ids), ptr(psi), ptr(c_s), ptr(hyp), C.c_float(w_ctc), nutt, beam, P, V, ptr(c_local),
                                      ptr(top_s), ptr(top_i), stream_ptr()), "eamd_beam_select")
    return top_s, top_i, c_local


def topk_rows(x, k, idx32=False):
    """(values [rows, k], indices [rows, k] int64) of the k largest of each row of a contiguous fp32 [rows, n] tensor, sorted
    (value descending, ties by ascending index): one launch, graph-replay safe (torch.topk's multi-block path is neither).
    idx32: a third result, the indices as int32 (the same launch)"""
    rows, n = x.shape
    if x.dtype != torch.float32 or not x.is_contiguous() or k > 64:
        raise _lib.EamdError("topk_rows: contiguous float32 [rows, n], k <= 64")
    vals = torch.empty(rows, k, device=x.device, dtype=torch.float32)
    idx = torch.empty(rows, k, device=x.device, dtype=torch.int64)
    i32 = torch.empty(rows, k, device=x.device, dtype=torch.int32) if idx32 else None
    check(_lib.lib().eamd_topk_rows_i32(ptr(x), C.c_int64(n), rows, n, k, ptr(vals), ptr(idx), ptr(i32), stream_ptr()), "eamd_topk_rows")
    return (vals, idx, i32) if idx32 else (vals, idx)


def beam_step(pre, ids, psi, c_s, hyp, w_ctc, nutt, beam, L, step, eos, maxlen, sc_in, logps, yseq_in, dyn=None, slot_in=None):
    """selection and bookkeeping of a BeamSearch step with a pre-beam in one launch (eamd_beam_step = eamd_beam_select +
    eamd_beam_finish) -> (sc_out [ns, n], yseq_out, hyp_out, hyp_i, tok_i, tok32, cs_out, rec)"""
    n, V = pre.shape
    P = ids.shape[1]
    ns, W = sc_in.shape[0], yseq_in.shape[1]
    nf = len(logps)
    dev = pre.device
    for t_ in (pre, psi, c_s, hyp, sc_in) + tuple(logps):
        if t_.dtype != torch.float32 or not t_.is_contiguous():
            raise _lib.EamdError("beam_step: contiguous float32 score tensors")
    for t_ in (ids, maxlen, yseq_in):
        if t_.dtype != torch.int64 or not t_.is_contiguous():
            raise _lib.EamdError("beam_step: contiguous int64 index tensors")
    assert n == nutt * beam and psi.shape == (n, P) and c_s.numel() == n and hyp.numel() == n and sc_in.shape[1] == n
    assert yseq_in.shape[0] == n and all(lp.shape == (n, V) for lp in logps) and ns == nf + 1 and maxlen.numel() == nutt
    c_local = torch.empty(n, P, device=dev, dtype=torch.float32)
    sc_out = torch.empty(ns, n, device=dev, dtype=torch.float32)
    yseq_out = torch.empty(n, W, device=dev, dtype=torch.int64)
    hyp_out, cs_out = (torch.empty(n, device=dev, dtype=torch.float32) for _ in range(2))
    hyp_i, tok_i = (torch.empty(n, device=dev, dtype=torch.int64) for _ in range(2))
    tok32 = torch.empty(n, device=dev, dtype=torch.int32)
    # dyn = (step_dev, step_out, ring): the step index read from the device, step + 1 written to step_out, the log row into slot
    # step % R of the ring [R, n, 3 + ns + W] (one graph for every step)
    step_dev = step_out = None
    ring = 0
    if dyn is not None:
        step_dev, step_out, rec = dyn
        ring = rec.shape[0]
        assert rec.shape[1:] == (n, 3 + ns + W) and rec.is_contiguous() and step_dev.dtype == torch.int32 and step_out.dtype == torch.int32
    else:
        rec = torch.empty(n, 3 + ns + W, device=dev, dtype=torch.float32)
    slot_out, Lcap = None, 0
    if slot_in is not None:
        assert slot_in.dtype == torch.int32 and slot_in.is_contiguous() and slot_in.shape[0] == n
        slot_out, Lcap = torch.empty_like(slot_in), slot_in.shape[1]
    arr = (C.c_void_p * 4)(*[lp.data_ptr() for lp in logps] + [None] * (4 - nf))
    check(_lib.lib().eamd_beam_step_dyn(ptr(pre), ptr(ids), ptr(psi), ptr(c_s), ptr(hyp), C.c_float(w_ctc), nutt, beam, P, V, W, L, step, eos,
                                        ptr(maxlen), ns, nf, ptr(sc_in), arr, ptr(yseq_in), ptr(c_local), ptr(sc_out), ptr(yseq_out),
                                        ptr(hyp_out), ptr(hyp_i), ptr(tok_i), ptr(tok32), ptr(cs_out), ptr(rec), ptr(step_dev),
                                        ptr(step_out), ring, ptr(slot_in), ptr(slot_out), Lcap, stream_ptr()), "eamd_beam_step")
    if slot_in is not None:       # (a ninth result: the cached decoder's slot table behind the selection)
        return sc_out, yseq_out, hyp_out, hyp_i, tok_i, tok32, cs_out, rec, slot_out
    return sc_out, yseq_out, hyp_out, hyp_i, tok_i, tok32, cs_out, rec


def beam_finish(top_s, top_i, beam, V, L, step, eos, maxlen, sc_in, logps, c_local, full_mode, ids, yseq_in):
    """the bookkeeping of a beam step after the selection in one launch (eamd_beam_finish) ->
    (sc_out [ns, n], yseq_out, hyp_out, hyp_i, tok_i, pos, rec)"""
    n = top_s.numel()
    ns, W = sc_in.shape[0], yseq_in.shape[1]
    nf = len(logps)
    dev = top_s.device
    for t_ in (top_s, sc_in, c_local) + tuple(logps):
        if t_ is not None and (t_.dtype != torch.float32 or not t_.is_contiguous()):
            raise _lib.EamdError("beam_finish: contiguous float32 score tensors")
    for t_ in (top_i, maxlen, ids, yseq_in):
        if t_ is not None and (t_.dtype != torch.int64 or not t_.is_contiguous()):
            raise _lib.EamdError("beam_finish: contiguous int64 index tensors")
    assert top_i.numel() == n and sc_in.shape[1] == n and yseq_in.shape[0] == n and all(lp.shape == (n, V) for lp in logps)
    assert ns in (nf, nf + 1) and (ns == nf or (c_local is not None and c_local.shape[0] == n))
    sc_out = torch.empty(ns, n, device=dev, dtype=torch.float32)
    yseq_out = torch.empty(n, W, device=dev, dtype=torch.int64)
    hyp_out = torch.empty(n, device=dev, dtype=torch.float32)
    hyp_i, tok_i, pos = (torch.empty(n, device=dev, dtype=torch.int64) for _ in range(3))
    rec = torch.empty(n, 3 + ns + W, device=dev, dtype=torch.float32)
    arr = (C.c_void_p * 4)(*[lp.data_ptr() for lp in logps] + [None] * (4 - nf))
    check(_lib.lib().eamd_beam_finish(ptr(top_s), ptr(top_i), n, beam, V, W, L, step, eos, ptr(maxlen), ns, nf, ptr(sc_in), arr,
                                      ptr(c_local), C.c_int64(c_local.shape[1] if c_local is not None else 0), int(bool(full_mode)),
                                      ptr(ids), ids.shape[1] if ids is not None else 0, ptr(yseq_in), ptr(sc_out), ptr(yseq_out),
                                      ptr(hyp_out), ptr(hyp_i), ptr(tok_i), ptr(pos), ptr(rec), stream_ptr()), "eamd_beam_finish")
    return sc_out, yseq_out, hyp_out, hyp_i, tok_i, pos, rec


def log_softmax_rows(x):
    rows, V = x.shape
    y = torch.empty_like(x)
    check(_lib.lib().eamd_log_softmax_rows(ptr(x), ptr(y), rows, V, stream_ptr()), "eamd_log_softmax_rows")
    return y


# ---- element-wise ------------------------------------------------------------------------------
def axpby(x, y, a=1.0, b=1.0, out=None):
    if out is None:
        out = torch.empty_like(x)
    assert y is None or y.numel() == x.numel()
    check(_lib.lib().eamd_axpby(ptr(x), ptr(y), ptr(out), C.c_int64(x.numel()), C.c_float(a), C.c_float(b),
                                stream_ptr()), "eamd_axpby")
    return out


def scale_dev(x, scale_t, extra=1.0, out=None):
    if out is None:
        out = torch.empty_like(x)
    check(_lib.lib().eamd_scale_dev(ptr(x), ptr(scale_t), ptr(out), C.c_int64(x.numel()), C.c_float(extra),
                                    stream_ptr()), "eamd_scale_dev")
    return out


def glu_fwd(a, Cc):
    rows = a.shape[0]
    y = torch.empty(rows, Cc, device=a.device, dtype=torch.float32)
    check(_lib.lib().eamd_glu_fwd(ptr(a), ptr(y), C.c_int64(rows), Cc, stream_ptr()), "eamd_glu_fwd")
    return y


def glu_bwd(dy, a, Cc, out_dtype=torch.float32):
    rows = a.shape[0]
    dx = torch.empty(a.shape, device=a.device, dtype=out_dtype)
    d32, d16 = (ptr(dx), None) if out_dtype == torch.float32 else (None, ptr(dx))
    check(_lib.lib().eamd_glu_bwd(ptr(dy), ptr(a), d32, d16, C.c_int64(rows), Cc, stream_ptr()), "eamd_glu_bwd")
    return dx


def add_bias2(q, u, v, rows=None, D=None, ldq=None, q_off=0):
    """qu = q + u, qv = q + v (dense [rows, D] outputs); q may be a column block (q_off, ldq) of a wider matrix"""
    if rows is None:
        rows, D = q.shape
    ldq = D if ldq is None else ldq
    assert u.numel() == D and v.numel() == D and q.numel() >= q_off + (rows - 1) * ldq + D
    qu = torch.empty(rows, D, device=q.device, dtype=q.dtype)
    qv = torch.empty(rows, D, device=q.device, dtype=q.dtype)
    check(_lib.lib().eamd_add_bias2(ptr(q, q_off), C.c_int64(ldq), ptr(u), ptr(v), ptr(qu), ptr(qv), C.c_int64(rows), D,
                                    1 if q.dtype == torch.bfloat16 else 0, stream_ptr()), "eamd_add_bias2")
    return qu, qv


def add_cast(a, b, out=None, out_off=0, ld_out=None):
    """bf16(a + b) from dense fp32 [rows, cols] inputs (b optional); optionally into a column block of `out`
    (an fp32 `out` takes the sum as it is)"""
    rows, cols = a.shape
    if out is None:
        out = torch.empty(a.shape, device=a.device, dtype=torch.bfloat16)
        ld_out = cols
    if out.dtype == torch.float32:
        assert out.numel() >= out_off + (rows - 1) * ld_out + cols
        check(_lib.lib().eamd_add_block_f32(ptr(a), ptr(b), ptr(out, out_off), C.c_int64(rows), cols, C.c_int64(ld_out),
                                            stream_ptr()), "eamd_add_block_f32")
        return out
    assert out.dtype == torch.bfloat16 and out.numel() >= out_off + (rows - 1) * ld_out + cols
    check(_lib.lib().eamd_add_cast_bf16(ptr(a), ptr(b), ptr(out, out_off), C.c_int64(rows), cols, C.c_int64(ld_out),
                                        stream_ptr()), "eamd_add_cast_bf16")
    return out


def add_cast_colsum2(a, b, suma, sumb, out=None, out_off=0, ld_out=None):
    """a + b in the operand dtype (bf16 in fast mode, fp32 otherwise; optionally into a column block of `out`) and
    suma += column sums of a, sumb += column sums of b in one pass; falls back to add_cast + two colsum launches for
    widths the fused kernel does not take"""
    rows, cols = a.shape
    if out is None:
        out = torch.empty(a.shape, device=a.device, dtype=act_dtype())
        ld_out = cols
    assert out.dtype in (torch.bfloat16, torch.float32) and out.numel() >= out_off + (rows - 1) * ld_out + cols
    assert a.is_contiguous() and b.is_contiguous() and suma.numel() == cols and sumb.numel() == cols
    L = _lib.lib()
    fn, name = ((L.eamd_add_cast_colsum2, "eamd_add_cast_colsum2") if out.dtype == torch.bfloat16
                else (L.eamd_add_colsum2_f32, "eamd_add_colsum2_f32"))
    rc = fn(ptr(a), ptr(b), ptr(out, out_off), C.c_int64(ld_out), ptr(suma), ptr(sumb), C.c_int64(rows), cols, stream_ptr())
    if rc == _lib.EAMD_EUNSUPPORTED:
        colsum(a, suma)
        colsum(b, sumb)
        return add_cast(a, b, out=out, out_off=out_off, ld_out=ld_out)
    check(rc, name)
    return out


def embed_pe(tok, table, pe, U, scale, pos_offset=0, pos_dev=None):
    """out[r] = table[tok[r]] * scale + pe[r % U + pos_offset]   (pe None: plain embedding lookup).  tok contiguous, or one
    column of a wider buffer ([n, 1] with a row stride: the newest tokens of a beam step's prefixes)"""
    rows = tok.numel()
    D = table.shape[1]
    assert tok.dtype == torch.int64
    ldt = 1
    if not tok.is_contiguous():
        assert tok.dim() == 2 and tok.shape[1] == 1 and tok.stride(0) >= 1
        ldt = tok.stride(0)
    assert pe is None or (pe.shape[0] >= U + pos_offset and pe.shape[1] == D)
    out = torch.empty(rows, D, device=table.device, dtype=torch.float32)
    # pos_dev: int32 device scalar added to pos_offset (the step index of a replayed beam step)
    check(_lib.lib().eamd_embed_pe_dyn(ptr(tok), C.c_int64(ldt), ptr(table), ptr(pe), ptr(out), C.c_int64(rows), U, D,
                                       C.c_float(scale), pos_offset, ptr(pos_dev), stream_ptr()), "eamd_embed_pe")
    return out


def embed_bwd(tok, dout, dtable, scale, pad_idx=-1):
    rows, D = dout.shape
    check(_lib.lib().eamd_embed_bwd(ptr(tok), ptr(dout), ptr(dtable), C.c_int64(rows), D, C.c_float(scale),
                                    C.c_int64(pad_idx), stream_ptr()), "eamd_embed_bwd")


def posenc(x, pe, T, scale):
    rows, D = x.shape
    assert pe.shape[0] >= T and pe.shape[1] == D
    out = torch.empty_like(x)
    check(_lib.lib().eamd_posenc(ptr(x), ptr(pe), ptr(out), C.c_int64(rows), T, D, C.c_float(scale), stream_ptr()),
          "eamd_posenc")
    return out


def posenc_scaled(x, pe, alpha, T, scale=1.0):
    rows, D = x.shape
    assert pe.shape[0] >= T and pe.shape[1] == D and alpha.numel() == 1
    out = torch.empty_like(x)
    check(_lib.lib().eamd_posenc_scaled(ptr(x), ptr(pe), ptr(alpha), ptr(out), C.c_int64(rows), T, D, C.c_float(scale),
                                        stream_ptr()), "eamd_posenc_scaled")
    return out


def posenc_scaled_bwd(dout, pe, dalpha, T):
    rows, D = dout.shape
    check(_lib.lib().eamd_posenc_scaled_bwd(ptr(dout), ptr(pe), ptr(dalpha), C.c_int64(rows), T, D, stream_ptr()),
          "eamd_posenc_scaled_bwd")


_rng = {"step": None, "salt": 0}


def rng_state(device):
    """device-resident step counter that seeds every dropout mask of the current training step"""
    st = _rng["step"]
    if st is None or st.device != torch.device(device):
        st = torch.zeros(1, device=device, dtype=torch.int64)
        _rng["step"] = st
    return st


def rng_advance(device):
    check(_lib.lib().eamd_rng_advance(ptr(rng_state(device)), stream_ptr()), "eamd_rng_advance")


def manual_seed(seed, device="cuda"):
    rng_state(device).fill_(int(seed))


def new_salt():
    """unique id for a dropout site (module instance x use); masks of different sites are independent"""
    _rng["salt"] += 1
    return _rng["salt"]


def dropout(x, p, salt, act=ACT_NONE, out_dtype=None):
    """y = act(x) * mask / (1-p); calling it again with the same salt in the same step re-applies the
    same mask (that is the backward pass).  x fp32 or bf16."""
    out_dtype = x.dtype if out_dtype is None else out_dtype
    y = torch.empty(x.shape, device=x.device, dtype=out_dtype)
    check(_lib.lib().eamd_dropout(ptr(x), ptr(y), C.c_int64(x.numel()), C.c_float(p), ptr(rng_state(x.device)),
                                  C.c_uint64(salt), act, 1 if x.dtype == torch.bfloat16 else 0,
                                  1 if out_dtype == torch.bfloat16 else 0, stream_ptr()), "eamd_dropout")
    return y


# ---- convolution module --------------------------------------------------------------------------
def dwconv_fwd(x, w, bias, B, T, Cc, K):
    assert x.numel() == B * T * Cc and w.numel() == Cc * K
    y = torch.empty_like(x)
    check(_lib.lib().eamd_dwconv_fwd(ptr(x), ptr(w), ptr(bias), ptr(y), B, T, Cc, K, stream_ptr()),
          "eamd_dwconv_fwd")
    return y


def dwconv_bwd_x(dy, w, B, T, Cc, K):
    dx = torch.empty_like(dy)
    check(_lib.lib().eamd_dwconv_bwd_x(ptr(dy), ptr(w), ptr(dx), B, T, Cc, K, stream_ptr()), "eamd_dwconv_bwd_x")
    return dx


def dwconv_bwd_w(dy, x, dw, db, B, T, Cc, K):
    assert dw.numel() == Cc * K and (db is None or db.numel() == Cc)
    check(_lib.lib().eamd_dwconv_bwd_w(ptr(dy), ptr(x), ptr(dw), ptr(db), B, T, Cc, K, stream_ptr()),
          "eamd_dwconv_bwd_w")


def dwconv_glu_fwd(a, w, bias, B, T, Cc, K, bn=None):
    """y [B*T, Cc] = dwconv(GLU(a)) for a [B*T, 2*Cc] (value | gate columns): GLU(a) is formed on load, never written.
    bn = (eps, momentum, running_mean, running_var, num_batches_tracked): training-mode BatchNorm statistics of y from
    the same launch (partials in its epilogue + eamd_bn_finalize) -> (y, mean, rstd).
    Returns None when the library declines (kernel size beyond the LDS-tiled kernel)."""
    assert a.numel() == B * T * 2 * Cc and w.numel() == Cc * K and a.dtype == torch.float32
    y = torch.empty(B * T, Cc, device=a.device, dtype=torch.float32)
    nslab = B * ((T + 63) // 64)
    part = torch.empty(3 * Cc * nslab, device=a.device, dtype=torch.float32) if bn is not None else None
    rc = _lib.lib().eamd_dwconv_glu_fwd(ptr(a), ptr(w), ptr(bias), ptr(y), ptr(part), B, T, Cc, K, stream_ptr())
    if rc == _lib.EAMD_EUNSUPPORTED:
        return None
    check(rc, "eamd_dwconv_glu_fwd")
    if bn is None:
        return y
    eps, momentum, running_mean, running_var, nbt = bn
    mean = torch.empty(Cc, device=a.device, dtype=torch.float32)
    rstd = torch.empty(Cc, device=a.device, dtype=torch.float32)
    check(_lib.lib().eamd_bn_finalize(ptr(part), nslab, ptr(mean), ptr(rstd), ptr(running_mean), ptr(running_var), ptr(nbt), Cc,
                                      C.c_float(eps), C.c_float(momentum), stream_ptr()), "eamd_bn_finalize")
    return y, mean, rstd


def dwconv_glu_bwd_x(dy, w, a, B, T, Cc, K, out_dtype=torch.float32):
    """da [B*T, 2*Cc] = GLU'(a) . dwconv_bwd_x(dy): the gradient of the GLU output is never written"""
    da = torch.empty(a.shape, device=a.device, dtype=out_dtype)
    check(_lib.lib().eamd_dwconv_glu_bwd_x(ptr(dy), ptr(w), ptr(a), ptr(da), int(out_dtype == torch.bfloat16), B, T, Cc, K,
                                           stream_ptr()), "eamd_dwconv_glu_bwd_x")
    return da


def dwconv_glu_bwd_w(dy, a, dw, db, B, T, Cc, K):
    assert dw.numel() == Cc * K and (db is None or db.numel() == Cc)
    check(_lib.lib().eamd_dwconv_glu_bwd_w(ptr(dy), ptr(a), ptr(dw), ptr(db), B, T, Cc, K, stream_ptr()),
          "eamd_dwconv_glu_bwd_w")


def bn_stats(x, M, Cc, eps, momentum, running_mean, running_var, num_batches_tracked=None, bound=None):
    """bound = (T, int32 device scalar): rows (b, t) with t >= bound[0] are left out (eamd_bn_stats_bounded)"""
    nslab = _lib.lib().eamd_bn_nslab(C.c_int64(M), Cc)
    ws = torch.empty(3 * Cc * nslab, device=x.device, dtype=torch.float32)
    mean = torch.empty(Cc, device=x.device, dtype=torch.float32)
    rstd = torch.empty(Cc, device=x.device, dtype=torch.float32)
    assert num_batches_tracked is None or (num_batches_tracked.dtype == torch.int64 and num_batches_tracked.is_cuda)
    if bound is not None:
        check(_lib.lib().eamd_bn_stats_bounded(ptr(x), ptr(ws), ptr(mean), ptr(rstd), ptr(running_mean), ptr(running_var),
                                               ptr(num_batches_tracked), C.c_int64(M), Cc, C.c_float(eps), C.c_float(momentum),
                                               int(bound[0]), ptr(bound[1]), stream_ptr()), "eamd_bn_stats_bounded")
        return mean, rstd
    check(_lib.lib().eamd_bn_stats(ptr(x), ptr(ws), ptr(mean), ptr(rstd), ptr(running_mean), ptr(running_var),
                                   ptr(num_batches_tracked), C.c_int64(M), Cc, C.c_float(eps), C.c_float(momentum), stream_ptr()),
          "eamd_bn_stats")
    return mean, rstd


def bn_apply(x, mean, rstd, gamma, beta, M, Cc, act, out_dtype=torch.float32):
    y = torch.empty(x.shape, device=x.device, dtype=out_dtype)
    check(_lib.lib().eamd_bn_apply(ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ptr(y), C.c_int64(M), Cc,
                                   act, 1 if out_dtype == torch.bfloat16 else 0, stream_ptr()), "eamd_bn_apply")
    return y


def bn_bwd(dy, x, mean, rstd, gamma, beta, dgamma, dbeta, M, Cc, act, training, bound=None):
    nslab = _lib.lib().eamd_bn_nslab(C.c_int64(M), Cc)
    ws = torch.empty((2 * nslab + 2) * Cc, device=x.device, dtype=torch.float32)
    dx = torch.empty_like(x)
    if bound is not None:
        check(_lib.lib().eamd_bn_bwd_bounded(ptr(dy), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ptr(ws), ptr(dx),
                                             ptr(dgamma), ptr(dbeta), C.c_int64(M), Cc, act, int(training), int(bound[0]),
                                             ptr(bound[1]), stream_ptr()), "eamd_bn_bwd_bounded")
        return dx
    check(_lib.lib().eamd_bn_bwd(ptr(dy), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ptr(ws), ptr(dx),
                                 ptr(dgamma), ptr(dbeta), C.c_int64(M), Cc, act, int(training), stream_ptr()),
          "eamd_bn_bwd")
    return dx


def conv1_fwd(x, w, bias, B, T, F, Cc, out_dtype=torch.float32):
    H, W = (T - 3) // 2 + 1, (F - 3) // 2 + 1
    assert x.numel() == B * T * F and w.numel() == Cc * 9
    y = torch.empty(B, H, W, Cc, device=x.device, dtype=out_dtype)
    check(_lib.lib().eamd_conv1_fwd(ptr(x), ptr(w), ptr(bias), ptr(y), B, T, F, Cc,
                                    1 if out_dtype == torch.bfloat16 else 0, stream_ptr()), "eamd_conv1_fwd")
    return y


def conv1_bwd_w(dy, x, dw, db, B, T, F, Cc):
    L = _lib.lib()
    ws = torch.empty(int(L.eamd_conv1_bwd_w_workspace(B, T, Cc)), device=x.device, dtype=torch.float32)
    check(L.eamd_conv1_bwd_w(ptr(dy), ptr(x), ptr(dw), ptr(db), ptr(ws), B, T, F, Cc,
                             1 if dy.dtype == torch.bfloat16 else 0, stream_ptr()), "eamd_conv1_bwd_w")


def conv2_weight_prep(w, out_dtype=torch.float32):
    Co, Ci = w.shape[0], w.shape[1]
    wf = torch.empty(9, Ci, Co, device=w.device, dtype=out_dtype)
    wd = torch.empty(9, Co, Ci, device=w.device, dtype=out_dtype)
    check(_lib.lib().eamd_conv2_weight_prep(ptr(w), ptr(wf), ptr(wd), Co, Ci,
                                            1 if out_dtype == torch.bfloat16 else 0, stream_ptr()),
          "eamd_conv2_weight_prep")
    return wf, wd


def conv2_weight_grad(dwf, dw, Co, Ci):
    check(_lib.lib().eamd_conv2_weight_grad(ptr(dwf), ptr(dw), Co, Ci, stream_ptr()), "eamd_conv2_weight_grad")


def permute4(src, dst, dims, dst_strides, accumulate=False):
    d0, d1, d2, d3 = dims
    s0, s1, s2, s3 = dst_strides
    assert src.numel() == d0 * d1 * d2 * d3
    assert (d0 - 1) * s0 + (d1 - 1) * s1 + (d2 - 1) * s2 + (d3 - 1) * s3 < dst.numel()
    check(_lib.lib().eamd_permute4(ptr(src), ptr(dst), d0, d1, d2, d3, C.c_int64(s0), C.c_int64(s1), C.c_int64(s2),
                                   C.c_int64(s3), int(accumulate), stream_ptr()), "eamd_permute4")


# ---- integer helpers ----------------------------------------------------------------------------
def add_sos_eos(ys_pad, sos, eos, ignore_id):
    B, L = ys_pad.shape
    assert ys_pad.dtype == torch.int64 and ys_pad.is_contiguous()
    ys_in = torch.empty(B, L + 1, device=ys_pad.device, dtype=torch.int64)
    ys_out = torch.empty(B, L + 1, device=ys_pad.device, dtype=torch.int64)
    olen = torch.empty(B, device=ys_pad.device, dtype=torch.int32)
    check(_lib.lib().eamd_add_sos_eos(ptr(ys_pad), ptr(ys_in), ptr(ys_out), ptr(olen), B, L, sos, eos, ignore_id,
                                      stream_ptr()), "eamd_add_sos_eos")
    return ys_in, ys_out, olen


def ctc_collapse(ids, hlens, blank):
    B, T = ids.shape
    assert ids.dtype == torch.int32 and ids.is_contiguous()
    out = torch.empty(B, T, device=ids.device, dtype=torch.int32)
    outlen = torch.empty(B, device=ids.device, dtype=torch.int32)
    check(_lib.lib().eamd_ctc_collapse(ptr(ids), ptr(hlens), ptr(out), ptr(outlen), B, T, blank, stream_ptr()),
          "eamd_ctc_collapse")
    return out, outlen


# ---- CTC -----------------------------------------------------------------------------------------
def ctc_loss(acts_btv, ys_pad, ilens, blank=0, ignore_id=-1, grad_scale=1.0, want_grad=True, time_major=False):
    """acts [B,T,V] raw activations (time_major: [T,B,V], warp-ctc's layout - read in place through the entry point's
    (stride_t, stride_b), the gradient comes back in the same layout); ys_pad [B,L] int64; ilens [B] int32 -> nll [B], grad"""
    if time_major:
        T, B, V = acts_btv.shape
        st, sb = B * V, V
    else:
        B, T, V = acts_btv.shape
        st, sb = V, T * V
    L = ys_pad.shape[1]
    assert acts_btv.is_contiguous() and ys_pad.is_contiguous() and ys_pad.dtype == torch.int64
    assert ilens.dtype == torch.int32 and ilens.numel() == B and ys_pad.shape[0] == B
    ws_bytes = _lib.lib().eamd_ctc_workspace_bytes(B, T, L)
    ws = torch.empty(ws_bytes, device=acts_btv.device, dtype=torch.uint8)
    nll = torch.empty(B, device=acts_btv.device, dtype=torch.float32)
    grad = torch.empty_like(acts_btv) if want_grad else None
    check(_lib.lib().eamd_ctc_loss(ptr(acts_btv), C.c_int64(st), C.c_int64(sb), ptr(ys_pad), ptr(ilens), ptr(nll),
                                   ptr(grad), C.c_int64(st), C.c_int64(sb), ptr(ws), B, T, V, L, blank, ignore_id,
                                   C.c_float(grad_scale), stream_ptr()), "eamd_ctc_loss")
    return nll, grad


def ctc_prefix_score(logp, r_prev, cand, last, olen, blank, eos):
    """logp [T,V] fp32; r_prev [nhyp,T,2]; cand [nhyp,ncand] int32; last, olen [nhyp] int32
    -> psi [nhyp,ncand], r_new [nhyp,ncand,T,2]"""
    T, V = logp.shape
    nhyp, ncand = cand.shape
    assert r_prev.shape == (nhyp, T, 2) and cand.dtype == torch.int32 and logp.is_contiguous()
    # (candidates come from top-k / arange over the V scores: in range by construction; checking would cost a host sync per beam step)
    psi = torch.empty(nhyp, ncand, device=logp.device, dtype=torch.float32)
    r_new = torch.empty(nhyp, ncand, T, 2, device=logp.device, dtype=torch.float32)
    check(_lib.lib().eamd_ctc_prefix_score(ptr(logp), ptr(r_prev.contiguous()), ptr(cand.contiguous()), ptr(last),
                                           ptr(olen), ptr(psi), ptr(r_new), nhyp, ncand, T, V, blank, eos,
                                           stream_ptr()), "eamd_ctc_prefix_score")
    return psi, r_new


def ctc_prefix_score_batch(logp, lens, per_utt, r_prev, cand, last, olen, blank, eos):
    """the hypotheses of several utterances at once: logp [U,Tmax,V], lens [U] int32 (device), r_prev [U*per_utt,Tmax,2],
    cand [U*per_utt,ncand] int32 -> psi [U*per_utt,ncand], r_new [U*per_utt,ncand,Tmax,2] (rows beyond an utterance's length: 0)"""
    U, Tmax, V = logp.shape
    nhyp, ncand = cand.shape
    assert nhyp == U * per_utt and r_prev.shape == (nhyp, Tmax, 2) and cand.dtype == torch.int32 and logp.is_contiguous()
    assert lens.dtype == torch.int32 and lens.numel() == U
    psi = torch.empty(nhyp, ncand, device=logp.device, dtype=torch.float32)
    r_new = torch.zeros(nhyp, ncand, Tmax, 2, device=logp.device, dtype=torch.float32)
    check(_lib.lib().eamd_ctc_prefix_score_batch(ptr(logp), ptr(lens), U, per_utt, ptr(r_prev.contiguous()), ptr(cand.contiguous()),
                                                 ptr(last), ptr(olen), ptr(psi), ptr(r_new), ncand, Tmax, V, blank, eos,
                                                 stream_ptr()), "eamd_ctc_prefix_score_batch")
    return psi, r_new


def ctc_prefix_psi(logp, lens, per_utt, r_prev, cand, last, olen, blank, eos, olen_dev=None):
    """log psi of the candidates as a parallel reduction over the frames (eamd_ctc_prefix_psi): logp [U, Tmax, V], r_prev
    [U * per_utt, Tmax, 2], cand [n, P] int32, last [n] int32, olen int -> psi [n, P]; None when the library declines (Tmax > 2048)"""
    U, Tmax, V = logp.shape
    nhyp, ncand = cand.shape
    assert nhyp == U * per_utt and r_prev.shape == (nhyp, Tmax, 2) and cand.dtype == torch.int32 and logp.is_contiguous()
    psi = torch.empty(nhyp, ncand, device=logp.device, dtype=torch.float32)
    # olen_dev: int32 device scalar added to olen (the step index of a replayed beam step)
    rc = _lib.lib().eamd_ctc_prefix_psi_dyn(ptr(logp), ptr(lens), U, per_utt, ptr(r_prev.contiguous()), ptr(cand.contiguous()), ptr(last),
                                            int(olen), ptr(olen_dev), ptr(psi), ncand, Tmax, V, blank, eos, stream_ptr())
    if rc == _lib.EAMD_EUNSUPPORTED:
        return None
    check(rc, "eamd_ctc_prefix_psi")
    return psi


def ctc_prefix_state(logp, lens, per_utt, r_prev, parent, tok, last, olen, alive, blank, out=None, olen_dev=None):
    """forward variables of the continuations that survived a selection (eamd_ctc_prefix_state): slot s continues hypothesis
    parent[s] (whose state is r_prev[parent[s]], last token last[parent[s]], prefix length olen + 1) with token tok[s];
    alive [n] = the slots' running scores (-inf: ended / empty) -> r [n, Tmax, 2]"""
    U, Tmax, V = logp.shape
    n = parent.numel()
    r = out if out is not None else torch.empty(n, Tmax, 2, device=logp.device, dtype=torch.float32)
    check(_lib.lib().eamd_ctc_prefix_state_dyn(ptr(logp), ptr(lens), U, per_utt, ptr(r_prev), ptr(parent), ptr(tok), ptr(last), int(olen),
                                               ptr(olen_dev), ptr(alive), ptr(r), Tmax, V, blank, stream_ptr()), "eamd_ctc_prefix_state")
    return r


# ---- optimizer -------------------------------------------------------------------------------------
def grad_norm(g, ws, out):
    check(_lib.lib().eamd_grad_norm(ptr(g), C.c_int64(g.numel()), ptr(ws), ptr(out), stream_ptr()),
          "eamd_grad_norm")


def sched_step(state, gnorm, mode, base_lr, factor, dmodel, warmup, beta1, beta2, max_norm):
    check(_lib.lib().eamd_sched_step(ptr(state), ptr(gnorm), mode, C.c_float(base_lr), C.c_float(factor),
                                     C.c_float(dmodel), C.c_float(warmup), C.c_float(beta1), C.c_float(beta2),
                                     C.c_float(max_norm), stream_ptr()), "eamd_sched_step")


def adam_step(p, g, m, v, state, beta1, beta2, eps, weight_decay, p16=None):
    n = p.numel()
    assert g.numel() == n and m.numel() == n and v.numel() == n and (p16 is None or p16.numel() == n)
    check(_lib.lib().eamd_adam_step(ptr(p), ptr(g), ptr(m), ptr(v), ptr(p16), C.c_int64(n), ptr(state), C.c_float(beta1),
                                    C.c_float(beta2), C.c_float(eps), C.c_float(weight_decay), stream_ptr()),
          "eamd_adam_step")


def adadelta_step(p, g, sq, acc, state, rho, weight_decay, p16=None):
    check(_lib.lib().eamd_adadelta_step(ptr(p), ptr(g), ptr(sq), ptr(acc), ptr(p16), C.c_int64(p.numel()), ptr(state),
                                        C.c_float(rho), C.c_float(weight_decay), stream_ptr()), "eamd_adadelta_step")


def add_gradient_noise(g, sigma, salt=0x6e6f697365):
    """g += sigma * N(0, 1) over a flat fp32 gradient buffer (draws keyed by the device step counter)"""
    check(_lib.lib().eamd_add_gradient_noise(ptr(g), C.c_int64(g.numel()), C.c_float(sigma), ptr(rng_state(g.device)),
                                             C.c_uint64(salt), stream_ptr()), "eamd_add_gradient_noise")


def make_gather(Cc, taps, Ho, Wo, Hin, Win, sh, sw):
    g = GatherT()
    g.enabled = 1
    g.C, g.ntap, g.Ho, g.Wo, g.Hin, g.Win, g.sh, g.sw = Cc, len(taps), Ho, Wo, Hin, Win, sh, sw
    for i, (dh, dw) in enumerate(taps):
        g.dh[i] = dh
        g.dw[i] = dw
    return g


def make_rowmap(Ho, Wo, Hc, Wc, sh, oh, sw, ow):
    m = RowMapT()
    m.enabled = 1
    m.Ho, m.Wo, m.Hc, m.Wc, m.sh, m.oh, m.sw, m.ow = Ho, Wo, Hc, Wc, sh, oh, sw, ow
    return m


# ---- recurrent layers / RNN-Transducer (rows a20, a21) -------------------------------------------
ACT_TANH, ACT_HARDTANH, ACT_SELU = 3, 4, 5      # eamd_act ids beyond relu / swish (nets_utils.py:485-498)


def act_fwd_any(x, act):
    y = torch.empty_like(x)
    check(_lib.lib().eamd_act_fwd(ptr(x), ptr(y), C.c_int64(x.numel()), act, stream_ptr()), "eamd_act_fwd")
    return y


def act_bwd_any(dy, x, act):
    dx = torch.empty_like(x)
    check(_lib.lib().eamd_act_bwd(ptr(dy), ptr(x), ptr(dx), C.c_int64(x.numel()), act, stream_ptr()), "eamd_act_bwd")
    return dx


def lstm_cell_fwd(gates, c_prev, h_prev, live, h, c, y, acts):
    B, H4 = gates.shape
    H = H4 // 4
    assert c_prev.numel() == B * H and h.numel() == B * H and c.numel() == B * H and acts.numel() == B * H4
    assert live is None or (live.dtype == torch.uint8 and live.numel() == B and h_prev is not None)
    check(_lib.lib().eamd_lstm_cell_fwd(ptr(gates), ptr(c_prev), ptr(h_prev), ptr(live), ptr(h), ptr(c), ptr(y),
                                        ptr(acts), B, H, stream_ptr()), "eamd_lstm_cell_fwd")


def lstm_cell_bwd(dy, dh, dc, acts, c_prev, c, live, dgates, dc_prev, dh_pass):
    B, H4 = acts.shape
    H = H4 // 4
    assert dgates.numel() == B * H4 and dc_prev.numel() == B * H
    check(_lib.lib().eamd_lstm_cell_bwd(ptr(dy), ptr(dh), ptr(dc), ptr(acts), ptr(c_prev), ptr(c), ptr(live),
                                        ptr(dgates), ptr(dc_prev), ptr(dh_pass), B, H, stream_ptr()),
          "eamd_lstm_cell_bwd")


def lstm_step_ok(B, H):
    """shapes the one-launch LSTM step kernels take"""
    return LSTM_FUSED_STEP and H % 64 == 0 and B <= 64


LSTM_FUSED_STEP = True      # tests flip this to compare against the GEMM + cell-kernel steps


# bench.py: list collecting the recurrent launches of one step as (tag, operand references, replay(stream_ptr), bytes)
_rnn_record = None


def lstm_step_fwd(gx_t, w_hh, b_hh, h_prev, c_prev, live_t, h, c, y, acts):
    B, H4 = gx_t.shape
    H = H4 // 4
    args = (ptr(gx_t), ptr(w_hh), ptr(b_hh), ptr(h_prev), ptr(c_prev), ptr(live_t), ptr(h), ptr(c), ptr(y), ptr(acts), B, H)
    fn = _lib.lib().eamd_lstm_step_fwd
    if _rnn_record is not None:
        _rnn_record.append(("lstm_step_fwd", (gx_t, w_hh, b_hh, h_prev, c_prev, live_t, h, c, y, acts),
                            lambda sp, args=args: check(fn(*args, sp), "eamd_lstm_step_fwd"), 4 * (H4 * H + 3 * B * H4)))
    check(fn(*args, stream_ptr()), "eamd_lstm_step_fwd")


def lstm_step_bwd(dy_t, dgates_next, w_t, dh_pass_in, dc, acts, c_prev, c, live_t, dgates, dc_prev, dh_pass):
    B, H4 = acts.shape
    H = H4 // 4
    args = (ptr(dy_t), ptr(dgates_next), ptr(w_t), ptr(dh_pass_in), ptr(dc), ptr(acts), ptr(c_prev), ptr(c), ptr(live_t),
            ptr(dgates), ptr(dc_prev), ptr(dh_pass), B, H)
    fn = _lib.lib().eamd_lstm_step_bwd
    if _rnn_record is not None:
        _rnn_record.append(("lstm_step_bwd", (dy_t, dgates_next, w_t, dh_pass_in, dc, acts, c_prev, c, live_t, dgates, dc_prev, dh_pass),
                            lambda sp, args=args: check(fn(*args, sp), "eamd_lstm_step_bwd"), 4 * (H4 * H + 4 * B * H4)))
    check(fn(*args, stream_ptr()), "eamd_lstm_step_bwd")


LSTM_PERSISTENT = True      # whole-sequence persistent LSTM launches (csrc/lstm_seq.hip); tests flip it
_lstm_seq_last_ws = None    # sync words of the most recent persistent launch (lstm_seq_status)


_comm_overlap = False      # a gradient all-reduce may run beside backward (train.GradReducer / GraphedDataParallelStep, world > 1)
_lstm_seq_sticky = {}       # device index -> int32 word: first give-up code of any persistent launch since it was last cleared


def set_comm_overlap(active):
    """the data-parallel drivers announce collectives that overlap backward: their kernels hold CUs, and a persistent LSTM launch
    needs one workgroup on EVERY CU at once (it would sit in its flag waits until the collective drains) - lstm_seq_ok then
    declines and the per-step kernels run"""
    global _comm_overlap
    _comm_overlap = bool(active)


def lstm_seq_ok(njobs, B, H):
    """shapes eamd_lstm_seq_fwd / _bwd take for `njobs` recurrences side by side (one workgroup per CU for the whole
    launch; the entry points themselves answer EAMD_EUNSUPPORTED for anything else)"""
    if not (LSTM_PERSISTENT and LSTM_FUSED_STEP and H % 64 == 0 and B <= 64 and H <= 1024) or _comm_overlap:
        return False
    cus = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
    return njobs * (H // 8) <= cus and njobs * (H // 16) * ((B + 15) // 16) <= cus


def _lstm_seq_ws(dev):
    global _lstm_seq_last_ws
    n = int(_lib.lib().eamd_lstm_seq_sync_bytes())
    _lstm_seq_last_ws = torch.empty(n, dtype=torch.uint8, device=dev)
    return _lstm_seq_last_ws


def lstm_seq_fwd(jobs, T, B, H):
    """jobs: list of (gx [T,B,4H], w_hh, b_hh, live [T,B] | None, h_out, c_out, y, acts, reverse) - ONE persistent launch"""
    arr = (_lib.LstmSeqFwdT * len(jobs))()
    for q, (gx, w, b, live, h, c, y, acts, rev) in zip(arr, jobs):
        q.gx, q.w_hh, q.b_hh, q.live = ptr(gx), ptr(w), ptr(b), ptr(live)
        q.h_out, q.c_out, q.y, q.acts, q.reverse = ptr(h), ptr(c), ptr(y), ptr(acts), int(bool(rev))
    ws = _lstm_seq_ws(jobs[0][0].device)
    fn = _lib.lib().eamd_lstm_seq_fwd
    args = (arr, len(jobs), T, B, H, ptr(ws))
    if _rnn_record is not None:
        _rnn_record.append(("lstm_seq_fwd", (jobs, ws, arr), lambda sp, args=args: check(fn(*args, sp), "eamd_lstm_seq_fwd"),
                            4 * len(jobs) * (4 * H * H + T * B * 9 * H), 2.0 * len(jobs) * (T - 1) * B * 4 * H * H))
    check(fn(*args, stream_ptr()), "eamd_lstm_seq_fwd")
    _lstm_seq_merge(ws)


def lstm_seq_bwd(jobs, T, B, H):
    """jobs: list of (dy [T,B,H], w_t [H,4H], acts, c_out, live | None, dgates (out), reverse = the FORWARD direction)"""
    arr = (_lib.LstmSeqBwdT * len(jobs))()
    for q, (dy, w_t, acts, c, live, dg, rev) in zip(arr, jobs):
        q.dy, q.w_t, q.acts, q.c_out, q.live, q.dgates, q.reverse = ptr(dy), ptr(w_t), ptr(acts), ptr(c), ptr(live), ptr(dg), int(bool(rev))
    ws = _lstm_seq_ws(jobs[0][2].device)
    fn = _lib.lib().eamd_lstm_seq_bwd
    args = (arr, len(jobs), T, B, H, ptr(ws))
    if _rnn_record is not None:
        _rnn_record.append(("lstm_seq_bwd", (jobs, ws, arr), lambda sp, args=args: check(fn(*args, sp), "eamd_lstm_seq_bwd"),
                            4 * len(jobs) * (4 * H * H + T * B * 11 * H), 2.0 * len(jobs) * (T - 1) * B * 4 * H * H))
    check(fn(*args, stream_ptr()), "eamd_lstm_seq_bwd")
    _lstm_seq_merge(ws)


def _lstm_seq_merge(ws):
    """fold the launch's status word into the device's sticky word (one tiny launch, no synchronisation)"""
    dev = ws.device
    st = _lstm_seq_sticky.get(dev.index)
    if st is None:
        if torch.cuda.is_current_stream_capturing():
            return                   # first persistent launch inside a capture: no persistent word yet (eager warm-ups come first)
        st = _lstm_seq_sticky[dev.index] = torch.zeros(1, dtype=torch.int32, device=dev)
    check(_lib.lib().eamd_lstm_seq_status_merge(ptr(ws), ptr(st), stream_ptr()), "eamd_lstm_seq_status_merge")


def lstm_seq_sticky_status(clear=True):
    """first give-up code of ANY persistent LSTM launch on the current device since the last clear (synchronises: call it
    where the host waits anyway - train.EpochRunner does once per epoch); 0 = all hand-off waits completed"""
    st = _lstm_seq_sticky.get(torch.cuda.current_device()) if torch.cuda.is_available() else None
    if st is None:
        return 0
    v = int(st.item())
    if v and clear:
        st.zero_()
    return v


def lstm_seq_status():
    """status word of the most recent persistent LSTM launch (synchronises): 0 = every hand-off wait completed"""
    if _lstm_seq_last_ws is None:
        return 0
    return int(_lib.lib().eamd_lstm_seq_status(ptr(_lstm_seq_last_ws), stream_ptr()))


def maxpool2x2_fwd(x):
    B, H, W, Cc = x.shape
    y = torch.empty(B, (H + 1) // 2, (W + 1) // 2, Cc, device=x.device, dtype=torch.float32)
    idx = torch.empty(y.shape, device=x.device, dtype=torch.uint8)
    check(_lib.lib().eamd_maxpool2x2_fwd(ptr(x), ptr(y), ptr(idx), B, H, W, Cc, stream_ptr()), "eamd_maxpool2x2_fwd")
    return y, idx


def maxpool2x2_bwd(dy, idx, shape):
    B, H, W, Cc = shape
    dx = torch.empty(B, H, W, Cc, device=dy.device, dtype=torch.float32)
    check(_lib.lib().eamd_maxpool2x2_bwd(ptr(dy), ptr(idx), ptr(dx), B, H, W, Cc, stream_ptr()), "eamd_maxpool2x2_bwd")
    return dx


def conv3x3_c1_fwd(x, w, bias, B, T, F, Cc, out_dtype=torch.float32):
    assert x.numel() == B * T * F and w.numel() == Cc * 9
    y = torch.empty(B, T, F, Cc, device=x.device, dtype=out_dtype)
    check(_lib.lib().eamd_conv3x3_c1_fwd(ptr(x), ptr(w), ptr(bias), ptr(y), B, T, F, Cc,
                                         1 if out_dtype == torch.bfloat16 else 0, stream_ptr()), "eamd_conv3x3_c1_fwd")
    return y


def conv3x3_c1_bwd_w(dy, x, dw, db, B, T, F, Cc):
    L = _lib.lib()
    ws = torch.empty(int(L.eamd_conv3x3_c1_bwd_w_workspace(B, T, Cc)), device=x.device, dtype=torch.float32)
    check(L.eamd_conv3x3_c1_bwd_w(ptr(dy), ptr(x), ptr(dw), ptr(db), ptr(ws), B, T, F, Cc,
                                  1 if dy.dtype == torch.bfloat16 else 0, stream_ptr()), "eamd_conv3x3_c1_bwd_w")


def joint_fwd(enc, dec, act, out_dtype=torch.float32):
    B, T, J = enc.shape
    U = dec.shape[1]
    assert dec.shape[0] == B and dec.shape[2] == J and enc.is_contiguous() and dec.is_contiguous()
    out = torch.empty(B, T, U, J, device=enc.device, dtype=out_dtype)
    o32, o16 = (ptr(out), None) if out_dtype == torch.float32 else (None, ptr(out))
    check(_lib.lib().eamd_joint_fwd(ptr(enc), ptr(dec), o32, o16, B, T, U, J, act, stream_ptr()), "eamd_joint_fwd")
    return out


def joint_bwd(dh, enc, dec, act):
    B, T, J = enc.shape
    U = dec.shape[1]
    assert dh.numel() == B * T * U * J and dh.dtype == torch.float32
    d_enc = torch.empty_like(enc)
    d_dec = torch.empty_like(dec)
    check(_lib.lib().eamd_joint_bwd(ptr(dh), ptr(enc), ptr(dec), ptr(d_enc), ptr(d_dec), B, T, U, J, act,
                                    stream_ptr()), "eamd_joint_bwd")
    return d_enc, d_dec


def rnnt_grad(logits, labels, tlens, ulens, blank, ws, gscale, scale, grad=None):
    """d loss / d logits from the lattice workspace `ws` of a previous rnnt_loss(..., return_ws=True) call;
    gscale: 0-dim / 1-element device tensor multiplied in (the upstream gradient of the scalar loss)"""
    B, T, U, V = logits.shape
    grad = torch.empty_like(logits) if grad is None else grad
    lab = labels if labels.numel() > 0 else torch.zeros(1, device=logits.device, dtype=torch.int32)
    check(_lib.lib().eamd_rnnt_grad(ptr(logits), ptr(lab), ptr(tlens), ptr(ulens), ptr(ws), ptr(grad), B, T, U, V, blank,
                                    ptr(gscale), C.c_float(scale), stream_ptr()), "eamd_rnnt_grad")
    return grad


def rnnt_loss(logits, labels, tlens, ulens, blank, grad=None, gscale=None, scale=1.0, return_ws=False):
    """logits [B,T,U,V] fp32; labels [B,U-1] int32; tlens/ulens [B] int32 -> per-utterance loss [B]"""
    B, T, U, V = logits.shape
    assert logits.dtype == torch.float32 and logits.is_contiguous()
    assert labels.dtype == torch.int32 and labels.is_contiguous() and labels.numel() == B * max(U - 1, 0)
    assert tlens.dtype == torch.int32 and ulens.dtype == torch.int32 and tlens.numel() == B and ulens.numel() == B
    assert grad is None or (grad.numel() == logits.numel() and grad.dtype == torch.float32)
    ws = torch.empty(int(_lib.lib().eamd_rnnt_workspace(B, T, U)), device=logits.device, dtype=torch.float32)
    loss = torch.empty(B, device=logits.device, dtype=torch.float32)
    lab = labels if labels.numel() > 0 else torch.zeros(1, device=logits.device, dtype=torch.int32)
    check(_lib.lib().eamd_rnnt_loss(ptr(logits), ptr(lab), ptr(tlens), ptr(ulens), ptr(ws), ptr(loss), ptr(grad),
                                    B, T, U, V, blank, ptr(gscale), C.c_float(scale), stream_ptr()), "eamd_rnnt_loss")
    return (loss, ws) if return_ws else loss


def rnnt_workspace(B, T, U, device):
    return torch.empty(int(_lib.lib().eamd_rnnt_workspace(B, T, U)), device=device, dtype=torch.float32)


def rnnt_node_stats(z_rows, labels, ws, node0, B, T, U, blank):
    """lse / log p(blank) / log p(label) of the lattice rows node0.. (z_rows [nrows, V] fp32) into the workspace"""
    nrows, V = z_rows.shape
    assert z_rows.dtype == torch.float32 and z_rows.is_contiguous() and labels.dtype == torch.int32
    check(_lib.lib().eamd_rnnt_node_stats(ptr(z_rows), ptr(labels), ptr(ws), C.c_int64(node0), C.c_int64(nrows), B, T, U, V,
                                          blank, stream_ptr()), "eamd_rnnt_node_stats")


RNNT_FUSED_STATS = True     # tests flip this to reach the stored-logits form (linear_fwd + eamd_rnnt_node_stats)


def rnnt_node_stats_fused(H2, W, b_out, col, ws, node0, B, T, U, blank):
    """the node statistics of the lattice rows node0.. straight from the joint activations H2 [nrows, J] (operand dtype) and
    the output projection W [V, J] / b_out: the logits GEMM runs with the row-statistics epilogue (no logits written) and
    eamd_rnnt_node_stats_part combines its partials.  col [nrows] int32: each node's next label or -1.
    Returns False if the library declines (the caller then stores the logits)."""
    if not RNNT_FUSED_STATS:
        return False
    nrows, J = H2.shape
    V = W.shape[0]
    tile = 128 if nrows >= 2048 else 64
    tn = (V + tile - 1) // tile
    dev = H2.device
    part = torch.empty(nrows * tn * 2, device=dev, dtype=torch.float32)
    zcol = torch.empty(nrows, device=dev, dtype=torch.float32)
    zfix = torch.empty(nrows, device=dev, dtype=torch.float32)
    try:
        gemm(H2, W, None, nrows, V, J, J, J, V, bias=b_out, epilogue=EPI_ROW_STATS, tile=tile,
             stats=(part, col, zcol, zfix, blank))
    except _lib.EamdError as e:
        if ("code %d" % _lib.EAMD_EUNSUPPORTED) in str(e):
            return False
        raise
    check(_lib.lib().eamd_rnnt_node_stats_part(ptr(part), ptr(zcol), ptr(zfix), ptr(ws), C.c_int64(node0), C.c_int64(nrows), tn,
                                               B, T, U, stream_ptr()), "eamd_rnnt_node_stats_part")
    return True


def rnnt_node_grad_fused(H2, W, b_out, labels, tlens, ulens, ws, node0, B, T, U, blank, gscale, scale, out_dtype):
    """d loss / d logits of the lattice rows node0.. -> [nrows, V] in out_dtype, written by the recomputing logits GEMM itself
    (epilogue 8 with eamd_rnnt_row_coef's per-node coefficients): no fp32 logits chunk, no separate gradient pass.
    Returns None if the library declines."""
    if not RNNT_FUSED_STATS:
        return None
    nrows, J = H2.shape
    V = W.shape[0]
    dev = H2.device
    rowc = torch.empty(nrows * 3, device=dev, dtype=torch.float32)
    col = torch.empty(nrows, device=dev, dtype=torch.int32)
    check(_lib.lib().eamd_rnnt_row_coef(ptr(labels), ptr(tlens), ptr(ulens), ptr(ws), ptr(rowc), ptr(col), C.c_int64(node0),
                                        C.c_int64(nrows), B, T, U, stream_ptr()), "eamd_rnnt_row_coef")
    dZ = torch.empty(nrows, V, device=dev, dtype=out_dtype)
    try:
        gemm(H2, W, dZ, nrows, V, J, J, J, V, bias=b_out, epilogue=EPI_ROW_GRAD, tile=128 if nrows >= 2048 else 64,
             stats=(rowc, col, blank, gscale, scale))
    except _lib.EamdError as e:
        if ("code %d" % _lib.EAMD_EUNSUPPORTED) in str(e):
            return None
        raise
    return dZ


def rnnt_alpha_beta(ws, tlens, ulens, B, T, U):
    loss = torch.empty(B, device=ws.device, dtype=torch.float32)
    check(_lib.lib().eamd_rnnt_alpha_beta(ptr(ws), ptr(tlens), ptr(ulens), ptr(loss), B, T, U, stream_ptr()),
          "eamd_rnnt_alpha_beta")
    return loss


def rnnt_node_grad(z_rows, labels, tlens, ulens, ws, node0, B, T, U, blank, gscale, scale, out_dtype=torch.float32):
    """d loss / d logits of the lattice rows node0.. -> [nrows, V] in out_dtype (fp32: written over z_rows)"""
    nrows, V = z_rows.shape
    if out_dtype == torch.float32:
        g32, g16, out = ptr(z_rows), None, z_rows
    else:
        out = torch.empty(nrows, V, device=z_rows.device, dtype=torch.bfloat16)
        g32, g16 = None, ptr(out)
    check(_lib.lib().eamd_rnnt_node_grad(ptr(z_rows), g32, g16, ptr(labels), ptr(tlens), ptr(ulens), ptr(ws), C.c_int64(node0),
                                         C.c_int64(nrows), B, T, U, V, blank, ptr(gscale), C.c_float(scale), stream_ptr()),
          "eamd_rnnt_node_grad")
    return out


def attloc_fwd(att_prev, conv_w, w_att, pre_enc, dec_proj, gvec, gb, lens, enc_h, scaling):
    B, T, A = pre_enc.shape
    Cc, K = (conv_w.shape[0], conv_w.shape[-1]) if conv_w is not None else (0, 0)   # no conv: additive attention
    R = conv_w.shape[2] if (conv_w is not None and conv_w.dim() == 4) else 1          # rows of attention history
    E = enc_h.shape[2]
    dev = enc_h.device
    assert dec_proj.shape == (B, A) and gvec.numel() == A and lens.dtype == torch.int32 and lens.numel() == B
    assert conv_w is None or (att_prev.numel() == B * R * T and w_att.shape == (A, Cc) and conv_w.numel() == Cc * R * K)
    e = torch.empty(B, T, device=dev, dtype=torch.float32)
    th = torch.empty(B, T, A, device=dev, dtype=torch.float32)
    conv = torch.empty(B, T, Cc, device=dev, dtype=torch.float32) if Cc else None
    w = torch.empty(B, T, device=dev, dtype=torch.float32)
    ctx = torch.empty(B, E, device=dev, dtype=torch.float32)
    check(_lib.lib().eamd_attloc_fwd(ptr(att_prev), ptr(conv_w), ptr(w_att), ptr(pre_enc), ptr(dec_proj), ptr(gvec),
                                     ptr(gb), ptr(lens), ptr(enc_h), C.c_float(scaling), ptr(e), ptr(th), ptr(conv),
                                     ptr(w), ptr(ctx), B, T, A, Cc, K, R, E, stream_ptr()), "eamd_attloc_fwd")
    return ctx, w, th, conv


def attloc_bwd_energy(dctx, dw_ext, w, enc_h, th, gvec, scaling, dgvec, dgb):
    B, T, A = th.shape
    E = enc_h.shape[2]
    dev = enc_h.device
    de = torch.empty(B, T, device=dev, dtype=torch.float32)
    d_enc_h = torch.empty(B, T, E, device=dev, dtype=torch.float32)
    df = torch.empty(B, T, A, device=dev, dtype=torch.float32)
    d_dec = zeros(B, A, device=dev)
    check(_lib.lib().eamd_attloc_bwd_energy(ptr(dctx), ptr(dw_ext), ptr(w), ptr(enc_h), ptr(th), ptr(gvec),
                                            C.c_float(scaling), ptr(de), ptr(d_enc_h), ptr(df), ptr(dgvec), ptr(dgb),
                                            ptr(d_dec), B, T, A, E, stream_ptr()), "eamd_attloc_bwd_energy")
    return d_enc_h, df, d_dec


ATTLOC_FUSED_BWD = True     # tests flip this to reach the GEMM form of the two products over mlp_att's weight


def attloc_bwd_energy_conv(dctx, dw_ext, w, enc_h, th, gvec, scaling, conv, w_att, dgvec, dgb, dw_att, acc=None):
    """attloc_bwd_energy plus dconv = df @ W_att and dw_att += df^T conv from the same pass over th
    (eamd_attloc_bwd_energy_conv).  Returns (d_enc_h, df, d_dec, dconv), or None if the library declines the shape.
    acc: a dict shared by the decoder steps of one utterance batch - d_enc_h and df (gradients of step-invariant tensors)
    are then kept as ONE running sum in it (created by the first call, added to by the later ones) and returned as such."""
    B, T, A = th.shape
    E, Cc = enc_h.shape[2], conv.shape[2]
    if not ATTLOC_FUSED_BWD or Cc > 16 or A % 4 or A > 1024:
        return None
    dev = enc_h.device
    lib = _lib.lib()
    nws = int(lib.eamd_attloc_bwd_workspace(B, T, A, Cc)) // 4
    ws = torch.empty(nws, device=dev, dtype=torch.float32)
    de = torch.empty(B, T, device=dev, dtype=torch.float32)
    accumulate = acc is not None and "df" in acc
    if accumulate:
        d_enc_h, df = acc["deh"], acc["df"]
    else:
        d_enc_h = torch.empty(B, T, E, device=dev, dtype=torch.float32)
        df = torch.empty(B, T, A, device=dev, dtype=torch.float32)
    dconv = torch.empty(B, T, Cc, device=dev, dtype=torch.float32)
    d_dec = zeros(B, A, device=dev)
    rc = lib.eamd_attloc_bwd_energy_conv(ptr(dctx), ptr(dw_ext), ptr(w), ptr(enc_h), ptr(th), ptr(gvec), C.c_float(scaling),
                                         ptr(conv), ptr(w_att), ptr(de), ptr(d_enc_h), ptr(df), ptr(dconv), ptr(dgvec),
                                         ptr(dgb), ptr(d_dec), ptr(dw_att), ptr(ws), int(accumulate), B, T, A, Cc, E,
                                         stream_ptr())
    if rc == _lib.EAMD_EUNSUPPORTED:
        return None
    check(rc, "eamd_attloc_bwd_energy_conv")
    if acc is not None and not accumulate:
        acc["deh"], acc["df"] = d_enc_h, df
    return d_enc_h, df, d_dec, dconv


def attloc_bwd_conv(dconv, conv_w, att_prev, dconv_w):
    B, T, Cc = dconv.shape
    K = conv_w.shape[-1]
    R = conv_w.shape[2] if conv_w.dim() == 4 else 1
    d_prev = torch.empty(att_prev.shape, device=dconv.device, dtype=torch.float32)
    check(_lib.lib().eamd_attloc_bwd_conv(ptr(dconv), ptr(conv_w), ptr(att_prev), ptr(d_prev), ptr(dconv_w), B, T, Cc,
                                          K, R, stream_ptr()), "eamd_attloc_bwd_conv")
    return d_prev


def attloc_convmax_fwd(att_prev, conv_w):
    """att_prev [B,T], conv_w [C,1,1,K] -> pooled [B,C] = max_t relu(conv), idx [B,C] int32 (AttLocRec front end)"""
    B, T = att_prev.shape
    Cc, K = conv_w.shape[0], conv_w.shape[-1]
    pooled = torch.empty(B, Cc, device=att_prev.device, dtype=torch.float32)
    idx = torch.empty(B, Cc, device=att_prev.device, dtype=torch.int32)
    check(_lib.lib().eamd_attloc_convmax_fwd(ptr(att_prev), ptr(conv_w), ptr(pooled), ptr(idx), B, T, Cc, K, stream_ptr()),
          "eamd_attloc_convmax_fwd")
    return pooled, idx


def attloc_convmax_bwd(dpool, pooled, idx, att_prev, conv_w, dconv_w):
    B, T = att_prev.shape
    Cc, K = conv_w.shape[0], conv_w.shape[-1]
    d_prev = zeros(B, T, device=att_prev.device)
    check(_lib.lib().eamd_attloc_convmax_bwd(ptr(dpool), ptr(pooled), ptr(idx), ptr(att_prev), ptr(conv_w), ptr(d_prev),
                                             ptr(dconv_w), B, T, Cc, K, stream_ptr()), "eamd_attloc_convmax_bwd")
    return d_prev


def mask_rows(x, keep):
    """x [rows, D] fp32, keep [rows] uint8 -> rows with keep == 0 zeroed"""
    rows, D = x.shape
    assert keep.dtype == torch.uint8 and keep.numel() == rows and x.dtype == torch.float32
    y = torch.empty_like(x)
    check(_lib.lib().eamd_mask_rows(ptr(x), ptr(keep), ptr(y), C.c_int64(rows), D, stream_ptr()), "eamd_mask_rows")
    return y


def att_dot_fwd(k, q, v, lens, scaling):
    """dot-product attention step: k [B,T,A], q [B,A] (both tanh-activated), v [B,T,E] -> ctx [B,E], w [B,T]"""
    B, T, A = k.shape
    E = v.shape[2]
    dev = k.device
    e = torch.empty(B, T, device=dev, dtype=torch.float32)
    w = torch.empty(B, T, device=dev, dtype=torch.float32)
    ctx = torch.empty(B, E, device=dev, dtype=torch.float32)
    L = _lib.lib()
    check(L.eamd_att_dot_energy_fwd(ptr(k), ptr(q), ptr(lens), ptr(e), B, T, A, stream_ptr()), "eamd_att_dot_energy_fwd")
    check(L.eamd_att_ctx_fwd(ptr(e), ptr(v), C.c_float(scaling), ptr(w), ptr(ctx), B, T, E, stream_ptr()),
          "eamd_att_ctx_fwd")
    return ctx, w


def att_dot_bwd(dctx, dw_ext, w, k, q, v, scaling):
    B, T, A = k.shape
    E = v.shape[2]
    dev = k.device
    de = torch.empty(B, T, device=dev, dtype=torch.float32)
    d_v = torch.empty(B, T, E, device=dev, dtype=torch.float32)
    dk = torch.empty(B, T, A, device=dev, dtype=torch.float32)
    dq = zeros(B, A, device=dev)
    dsum = zeros(1, device=dev)
    L = _lib.lib()
    check(L.eamd_att_ctx_bwd(ptr(dctx), ptr(dw_ext), ptr(w), ptr(v), C.c_float(scaling), ptr(de), ptr(d_v), ptr(dsum),
                             B, T, E, stream_ptr()), "eamd_att_ctx_bwd")
    check(L.eamd_att_dot_energy_bwd(ptr(de), ptr(k), ptr(q), ptr(dk), ptr(dq), B, T, A, stream_ptr()),
          "eamd_att_dot_energy_bwd")
    return d_v, dk, dq


# ---- host constants on the device ----------------------------------------------------------------
# Length vectors, padding masks and padded label matrices are integer work the reference does on the host
# (nets_utils.py:64-176, rnn/decoders.py:167-190).  They are small and depend only on the batch's lengths / labels,
# so the device copies are cached by content: the eager warm-up step of a batch uploads them once and a later
# hipGraph capture / replay of the same batch finds them resident (no host-to-device copy inside a capture).
_h2d_cache = {}


def h2d_cached(tag, array, device):
    """device tensor holding `array` (numpy), cached by (tag, dtype, shape, bytes)"""
    import numpy as np
    a = np.ascontiguousarray(array)
    key = (tag, str(device), a.dtype.str, a.shape, a.tobytes())
    t = _h2d_cache.get(key)
    if t is None:
        if len(_h2d_cache) > 4096:
            _h2d_cache.clear()
        t = torch.from_numpy(a.copy()).to(device)
        _h2d_cache[key] = t
    return t


_side_streams = {}
TWO_STREAM_BIRNN = True     # the two directions of a bidirectional recurrent layer on two streams (tests flip it)


def side_stream(device, i=0):
    """a cached second stream per device (independent branches of a step: the reverse direction of a BLSTM layer)"""
    key = (str(device), i)
    st = _side_streams.get(key)
    if st is None:
        st = torch.cuda.Stream(device=device)
        _side_streams[key] = st
    return st


def h2d_async(t_cpu, device):
    """small host tensor -> device without making the host wait for the stream: a copy from PAGEABLE memory blocks the
    host until everything queued before it has run (a whole training step behind a graph replay); from a pinned
    staging copy it is just enqueued (the caching host allocator keeps the staging block until the copy has run)"""
    if not t_cpu.is_cuda and torch.device(device).type == "cuda":
        return t_cpu.contiguous().pin_memory().to(device, non_blocking=True)
    return t_cpu.to(device)


def gru_cell_fwd(gx, gh, h_prev, live, h, y, acts):
    B, H3 = gx.shape
    H = H3 // 3
    assert gh.numel() == B * H3 and h_prev.numel() == B * H and h.numel() == B * H and acts.numel() == B * 4 * H
    check(_lib.lib().eamd_gru_cell_fwd(ptr(gx), ptr(gh), ptr(h_prev), ptr(live), ptr(h), ptr(y), ptr(acts), B, H,
                                       stream_ptr()), "eamd_gru_cell_fwd")


def gru_cell_bwd(dy, dh, acts, h_prev, live, dgx, dgh, dh_direct):
    B, H4 = acts.shape
    H = H4 // 4
    assert dgx.numel() == B * 3 * H and dgh.numel() == B * 3 * H and dh_direct.numel() == B * H
    check(_lib.lib().eamd_gru_cell_bwd(ptr(dy), ptr(dh), ptr(acts), ptr(h_prev), ptr(live), ptr(dgx), ptr(dgh),
                                       ptr(dh_direct), B, H, stream_ptr()), "eamd_gru_cell_bwd")


# ---- feature-side layers --------------------------------------------------------------------------
def specaug(x, lens=None, center=None, warped=None, fpos=None, flen=None, tpos=None, tlen=None):
    """x [B,T,F] fp32 -> augmented copy; all parameter arrays are int32 device tensors (see include/espnet_amd.h)"""
    B, T, F = x.shape
    assert x.dtype == torch.float32 and x.is_contiguous()
    y = torch.empty_like(x)
    nf = fpos.shape[1] if fpos is not None else 0
    nt = tpos.shape[1] if tpos is not None else 0
    check(_lib.lib().eamd_specaug(ptr(x), ptr(y), ptr(lens), ptr(center), ptr(warped), ptr(fpos), ptr(flen), nf, ptr(tpos),
                                  ptr(tlen), nt, B, T, F, stream_ptr()), "eamd_specaug")
    return y


def global_mvn(x, lens, mean, std):
    B, T, F = x.shape
    y = torch.empty_like(x)
    check(_lib.lib().eamd_global_mvn(ptr(x), ptr(y), ptr(lens), ptr(mean), ptr(std), B, T, F, stream_ptr()), "eamd_global_mvn")
    return y


def utterance_mvn(x, lens, norm_means, norm_vars, eps):
    B, T, F = x.shape
    y = torch.empty_like(x)
    ws = torch.empty(2 * B * F, device=x.device, dtype=torch.float32)
    check(_lib.lib().eamd_utterance_mvn(ptr(x), ptr(y), ptr(lens), ptr(ws), int(norm_means), int(norm_vars), C.c_float(eps),
                                        B, T, F, stream_ptr()), "eamd_utterance_mvn")
    return y


def reflect_pad(x, pad, ldy, tail):
    """x [B, L] fp32 -> flat buffer of B rows of stride ldy (+ `tail` zero floats): reflect padding by `pad` on both
    sides of every row (torch.stft center=True), zeros up to ldy"""
    B, L = x.shape
    assert x.dtype == torch.float32 and x.is_contiguous() and ldy >= L + 2 * pad
    y = torch.zeros(B * ldy + tail, device=x.device, dtype=torch.float32)
    check(_lib.lib().eamd_reflect_pad(ptr(x), C.c_int64(L), ptr(y), C.c_int64(ldy), B, L, pad, stream_ptr()),
          "eamd_reflect_pad")
    return y


def logmel(spec, ld, rows_per_utt, melmat, lo, hi, flens, B, T, F, log_scale=1.0, power_input=False):
    """-> [B, T, M] log-mel features (see include/espnet_amd.h: eamd_logmel)"""
    M = melmat.shape[1]
    assert melmat.shape[0] == F and melmat.is_contiguous() and lo.dtype == torch.int32 and hi.dtype == torch.int32
    need = ((B - 1) * rows_per_utt + T - 1) * ld + (F if power_input else 2 * F)
    if spec.numel() < need:
        raise _lib.EamdError(f"logmel: spectrum buffer too small ({spec.numel()} < {need})")
    out = torch.empty(B, T, M, device=spec.device, dtype=torch.float32)
    check(_lib.lib().eamd_logmel(ptr(spec), C.c_int64(ld), C.c_int64(rows_per_utt), ptr(melmat), ptr(lo), ptr(hi),
                                 ptr(flens) if flens is not None else None, ptr(out), B, T, F, M, C.c_float(log_scale),
                                 int(power_input), stream_ptr()), "eamd_logmel")
    return out


def unfold1d(x, B, T, Cc, k):
    """x [B*T, C] (fp32 / bf16) -> col [B*T, k*C]: rows t-(k-1)/2 .. t+(k-1)/2 of the same sequence side by side"""
    assert x.numel() == B * T * Cc and x.is_contiguous() and x.dtype in (torch.float32, torch.bfloat16)
    col = torch.empty(B * T, k * Cc, device=x.device, dtype=x.dtype)
    check(_lib.lib().eamd_unfold1d(ptr(x), ptr(col), B, T, Cc, k, 1 if x.dtype == torch.bfloat16 else 0, stream_ptr()),
          "eamd_unfold1d")
    return col


def fold1d(dcol, B, T, Cc, k):
    """adjoint of unfold1d on fp32: dcol [B*T, k*C] -> dx [B*T, C]"""
    assert dcol.numel() == B * T * k * Cc and dcol.dtype == torch.float32 and dcol.is_contiguous()
    dx = torch.empty(B * T, Cc, device=dcol.device, dtype=torch.float32)
    check(_lib.lib().eamd_fold1d(ptr(dcol), ptr(dx), B, T, Cc, k, stream_ptr()), "eamd_fold1d")
    return dx
