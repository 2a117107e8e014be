"""hipGraph capture with a node audit.

Why: on this ROCm (7.2, gfx950) a MEMSET node of a captured graph fills its block correctly on the first launch of the
instantiated graph and leaves half of it non-zero from the second launch on (tools/graph_node_census.py part 2: a 10-line probe
around hipMemsetAsync; profiles/r04_graph_census.txt).  torch ops that zero scratch with cudaMemsetAsync (torch.topk's
multi-block path, sort / scan temporaries) therefore compute on garbage counters in every replay but the first - the cause of the
round-3 "second replay" GPU memory access fault of the multi-utterance beam-step graphs and of the flag block finding in
csrc/lstm_seq.hip.  Our own kernels never use hipMemsetAsync (csrc/common.h: eamd_zero_async); this module makes sure nothing else
brings a memset node into a graph we replay: every capture site of the package creates its graph with `new_graph()` and calls
`audit()` on it, which walks the captured nodes (hipGraphGetNodes / hipGraphNodeGetType) and raises if one is a memset node.
"""
import ctypes as C

import torch

from . import _lib

NODE_KINDS = {0: "kernel", 1: "memcpy", 2: "memset", 3: "host", 4: "graph", 5: "empty", 6: "wait_event", 7: "event_record",
              8: "sem_signal", 9: "sem_wait", 10: "mem_alloc", 11: "mem_free", 12: "memcpy_from_symbol", 13: "memcpy_to_symbol",
              14: "batch_mem_op"}


class _MemsetParams(C.Structure):         # hip_runtime_api.h: hipMemsetParams
    _fields_ = [("dst", C.c_void_p), ("elementSize", C.c_uint), ("height", C.c_size_t), ("pitch", C.c_size_t),
                ("value", C.c_uint), ("width", C.c_size_t)]


_hip = None


def _runtime():
    global _hip
    if _hip is None:
        h = C.CDLL("libamdhip64.so")          # the runtime torch has already loaded
        h.hipGraphGetNodes.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        h.hipGraphNodeGetType.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        h.hipGraphMemsetNodeGetParams.argtypes = [C.c_void_p, C.POINTER(_MemsetParams)]
        h.hipGraphChildGraphNodeGetGraph.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        h.hipStreamGetCaptureInfo.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_ulonglong)]
        h.hipStreamGetCaptureInfo.restype = C.c_int
        for f in (h.hipGraphGetNodes, h.hipGraphNodeGetType, h.hipGraphMemsetNodeGetParams, h.hipGraphChildGraphNodeGetGraph):
            f.restype = C.c_int
        _hip = h
    return _hip


def capture_id(stream=None):
    """unique id of the capture the stream is in, 0 when it is not capturing (hipStreamGetCaptureInfo)"""
    h = _runtime()
    st = C.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)
    status, cid = C.c_int(0), C.c_ulonglong(0)
    if h.hipStreamGetCaptureInfo(st, C.byref(status), C.byref(cid)) != 0:
        return -1
    return int(cid.value) if status.value == 1 else 0          # hipStreamCaptureStatusActive = 1


def new_graph():
    """a CUDAGraph that keeps its hipGraph_t after the capture (instantiated at the first replay), so that audit() can walk it"""
    return torch.cuda.CUDAGraph(keep_graph=True)


def _walk(raw, kinds, memsets):
    h = _runtime()
    n = C.c_size_t(0)
    if h.hipGraphGetNodes(C.c_void_p(raw), None, C.byref(n)) != 0:
        raise _lib.EamdError("hipGraphGetNodes failed")
    if n.value == 0:
        return
    nodes = (C.c_void_p * n.value)()
    if h.hipGraphGetNodes(C.c_void_p(raw), nodes, C.byref(n)) != 0:
        raise _lib.EamdError("hipGraphGetNodes failed")
    for node in nodes:
        t = C.c_int(-1)
        if h.hipGraphNodeGetType(C.c_void_p(node), C.byref(t)) != 0:
            raise _lib.EamdError("hipGraphNodeGetType failed")
        name = NODE_KINDS.get(t.value, "type%d" % t.value)
        kinds[name] = kinds.get(name, 0) + 1
        if t.value == 2:
            p = _MemsetParams()
            if h.hipGraphMemsetNodeGetParams(C.c_void_p(node), C.byref(p)) == 0:
                memsets.append(dict(dst=int(p.dst or 0), element_size=int(p.elementSize), width=int(p.width), height=int(p.height),
                                    value=int(p.value)))
            else:
                memsets.append(dict(dst=0, element_size=0, width=0, height=0, value=0))
        elif t.value == 4:
            child = C.c_void_p(0)
            if h.hipGraphChildGraphNodeGetGraph(C.c_void_p(node), C.byref(child)) == 0 and child.value:
                _walk(child.value, kinds, memsets)


def node_census(g):
    """-> (dict kind -> count, list of memset-node parameters) of a graph captured into a new_graph() object"""
    raw = g.raw_cuda_graph()
    raw = int(raw) if not isinstance(raw, int) else raw
    kinds, memsets = {}, []
    _walk(raw, kinds, memsets)
    return kinds, memsets


def audit(g, what="captured graph"):
    """raises EamdError when the graph holds a memset node (see the module docstring); returns the node census otherwise"""
    kinds, memsets = node_census(g)
    if memsets:
        desc = ", ".join("%d x %d B at 0x%x" % (m["height"] * max(m["width"], 1), m["element_size"], m["dst"]) for m in memsets[:4])
        raise _lib.EamdError("%s holds %d memset node(s) (%s): they replay wrongly on this ROCm - zero the buffer with a kernel "
                             "(tensor.zero_() is one) or keep the op out of the capture" % (what, len(memsets), desc))
    return kinds
