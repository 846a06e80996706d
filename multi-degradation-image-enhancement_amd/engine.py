"""Host side of the HIP engine: checkpoint packing, workspace ownership, launches.

PyTorch is used here only for device memory and streams; every numeric
operation of the path runs in libmdie_hip.so.
"""
import ctypes as C
import os
import threading
import time

import numpy as np
import torch

from . import lib as L

DTYPES = {"fp32": L.F32, "f32": L.F32, "float32": L.F32, "bf16": L.BF16, "bfloat16": L.BF16,
          "fp16": L.F16, "f16": L.F16, "float16": L.F16, "half": L.F16}
TORCH_DTYPE = {L.F32: torch.float32, L.BF16: torch.bfloat16, L.F16: torch.float16}


def dtype_id(name):
    try:
        return DTYPES[str(name).lower()]
    except KeyError:
        raise ValueError(f"unknown precision {name!r}; use 'fp32', 'bf16' or 'fp16'") from None


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream_ptr(device):
    """the current stream of `device` as a hipStream_t (torch's raw-stream getter: 0.2 us instead of the 4 us a Stream object costs --
    an eager training step asks ~700 times)"""
    if _raw_stream is not None:
        idx = device.index if isinstance(device, torch.device) else torch.device(device).index
        return C.c_void_p(_raw_stream(torch.cuda.current_device() if idx is None else idx))
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _require_gpu(x, what):
    if not x.is_cuda:
        raise L.MdieError(f"{what}: tensor is on {x.device}; this path runs on the MI355X only (no CPU fallback)")


def pack_checkpoint(state_dict, dtype):
    """state_dict (any device, the reference's 236 keys) -> packed parameter blob (CPU uint8 tensor)."""
    nbytes = L.lib.mdie_cdan_param_bytes(dtype)
    blob = torch.zeros(nbytes, dtype=torch.uint8)
    keep, entries = [], []
    for name, value in state_dict.items():
        if not torch.is_tensor(value) or not value.dtype.is_floating_point:
            continue  # num_batches_tracked counters are not needed in eval
        a = np.ascontiguousarray(value.detach().to("cpu", torch.float32).numpy())
        keep.append(a)
        entries.append(L.Tensor(name.encode(), a.ctypes.data, a.size))
    arr = (L.Tensor * len(entries))(*entries)
    L.check(L.lib.mdie_cdan_pack_params(dtype, arr, len(entries), blob.data_ptr(), nbytes), "mdie_cdan_pack_params")
    return blob


# How encoder.conv4 treats its CUs -- three BIT-IDENTICAL forms (include/mdie.h: mdie_conv_desc.share_cu):
#   0  conv_wide, one persistent workgroup per CU: fastest alone, but it holds every CU (and all of its LDS) until the layer ends, and the
#      DenseBlock branches that run beside the layer queue behind it;
#   2  conv_wide in two shorter runs per CU: every CU goes back to the dispatcher half way (MDIE_FWD_YIELD_CU_CONV4);
#   1  conv_kernel: three ~50 KB workgroups per CU, 9 us slower alone (MDIE_FWD_SHARE_CU_CONV4);
#   3  form 2 with the dense1 branch started behind decoder.conv1 instead of behind encoder.conv4 (MDIE_FWD_LATE_DENSE1: 1-4 us on every
#      box where it was swept beside form 2).
# Which makes the STEP fastest depends on the box -- round 5, tools/sched_sweep.py, profiles/r05m_sched_sweep.txt, r05u_*: form 2 -29 ... -34 us
# of 1045 on two boxes, form 1 -30 us of 1052 on a third and +10 of 1007 on the fastest one; form 3 beat form 2 by 1-4 us on every box swept
# and costs the fastest boxes <= 0.4 %.  So forward() takes form 3 STATICALLY wherever conv_wide runs the layer (it never times anything and
# never synchronises: the C ABI's contract), and CdanEngine.tune() is an EXPLICIT call -- bench.py makes it in its untimed warm-up, a serving
# loop may -- that times the four forms once per (device, element type, batch shape) and remembers the winner for later forwards of that
# shape.  MDIE_SHARE_CU_CONV4 = 0 | 1 | 2 | 3 fixes the form for the process (then tune() returns it untimed).
_SHARE_CU = {}
_SHARE_CU_LOCK = threading.Lock()
DEFAULT_FORM = 3


def _share_cu_eligible(dtype, B, H, W):
    """conv_wide takes encoder.conv4 only for 16-bit types, maps of whole 32x16 tiles and at least 96 items (csrc/conv_wide.hip)"""
    h, w = H // 8, W // 8
    return dtype != L.F32 and H % 8 == 0 and W % 8 == 0 and h % 16 == 0 and w % 32 == 0 and B * (h // 16) * (w // 32) * 8 >= 96


class CdanEngine:
    """Eval-mode CDAN forward on one GPU.  One instance per (device, precision) AND per forward in flight: the engine owns
    one workspace and one set of side streams, so two forwards that may overlap (issued on different streams) need two
    engines (tools/bench_inflight.py, tools/bench_e2e.py); calls on one stream are ordered by the stream."""

    def __init__(self, device, precision="fp32"):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise L.MdieError(f"CdanEngine needs a GPU device, got {self.device} (no CPU fallback)")
        self.dtype = dtype_id(precision)
        self.params = None
        self._ws = None
        self._ws_key = None
        self.use_side_streams = os.environ.get("MDIE_SIDE_STREAMS", "1") != "0"
        self.block_tail = os.environ.get("MDIE_BLOCK_TAIL", "0") == "1"   # A/B runs: decoder.final_dense as ONE launch (bit-identical; slower than the chain: csrc/final_block.hip)
        mode = os.environ.get("MDIE_SHARE_CU_CONV4", "auto")
        self.share_cu = None if mode == "auto" else int(mode)          # None: the static default, or what an explicit tune() found; 0 .. 3: the form
        self._aux = C.c_void_p(0)
        with torch.cuda.device(self.device):
            L.check(L.lib.mdie_aux_create(C.byref(self._aux)), "mdie_aux_create")

    def _key(self, B, H, W):
        return (self.device.index, self.dtype, B, H, W, self.use_side_streams)

    def form(self, B, H, W):
        """the form forward() runs a batch of this shape in: the environment's, what an explicit tune() found, or the static default -- no timing, no sync"""
        if self.share_cu is not None:
            return self.share_cu
        got = _SHARE_CU.get(self._key(B, H, W))
        if got is not None:
            return got
        return DEFAULT_FORM if (self.use_side_streams and _share_cu_eligible(self.dtype, B, H, W)) else 0

    def tune(self, x, rounds=4, steps=40):
        """EXPLICIT, never called by forward(): decide, for x's batch shape, which of the bit-identical forms encoder.conv4 and the dense1 branch
        run in (above): `rounds` alternating rounds of `steps` eager forwards per form, the first round discarded (about half a second at
        B = 32, 256x256; SYNCHRONISES the device: call it from a warm-up, with nothing else in flight).  The runs must be LONG: what separates
        the forms is how the chip behaves under sustained load -- rounds of 8 forwards picked the wrong one on a box where 50-step runs
        differ by 1.6 % the other way (gpurun_out/r05o).  Returns the form (0 ... 3) and remembers it for later forwards of this shape;
        under stream capture, with a form fixed by the environment, or for a shape conv_wide does not take, nothing is timed."""
        B, _, H, W = x.shape
        key = self._key(B, H, W)
        if self.share_cu is not None or torch.cuda.is_current_stream_capturing():
            return self.form(B, H, W)
        if key in _SHARE_CU:
            return _SHARE_CU[key]
        if not self.use_side_streams or not _share_cu_eligible(self.dtype, B, H, W):
            return 0
        y = torch.empty_like(x, dtype=torch.float32)
        forms = (0, 2, 3, 1)
        times = {f: [] for f in forms}
        try:
            with torch.no_grad():
                for f in forms:
                    self.share_cu = f
                    for _ in range(3):
                        self.forward(x, out=y)
                for _ in range(rounds):
                    for f in forms:
                        self.share_cu = f
                        torch.cuda.synchronize(self.device)
                        t0 = time.perf_counter()
                        for _ in range(steps):
                            self.forward(x, out=y)
                        torch.cuda.synchronize(self.device)
                        times[f].append(time.perf_counter() - t0)
        finally:
            self.share_cu = None      # (also when a forward raised: a candidate form must not stay pinned)
        med = {f: sorted(t[1:])[(len(t) - 1) // 2] for f, t in times.items()}       # (median of the rounds behind the first)
        best = min(forms, key=lambda f: med[f])
        with _SHARE_CU_LOCK:
            _SHARE_CU[key] = best
        self.tuned = {"shape": (B, H, W), "form": best, "us_per_step": {"0 conv_wide, one run per CU": round(med[0] / steps * 1e6, 1),
                                                                         "2 conv_wide, two runs per CU": round(med[2] / steps * 1e6, 1),
                                                                         "3 as 2, dense1 behind dec.conv1": round(med[3] / steps * 1e6, 1),
                                                                         "1 conv_kernel": round(med[1] / steps * 1e6, 1)}}
        return best

    def __del__(self):
        try:
            if self._aux:
                L.lib.mdie_aux_destroy(self._aux)
                self._aux = C.c_void_p(0)
        except Exception:
            pass

    def load(self, state_dict):
        self.params = pack_checkpoint(state_dict, self.dtype).to(self.device)
        return self

    def _workspace(self, B, H, W):
        key = (B, H, W)
        if self._ws_key != key:
            n = L.lib.mdie_cdan_workspace_bytes(self.dtype, B, H, W)
            if n == 0:
                raise L.MdieError(f"unsupported input extent {B}x3x{H}x{W}: H and W must be multiples of 8")
            if self._ws is None or self._ws.numel() < n:   # a larger buffer serves smaller batches (routed groups) as is
                self._ws = None  # release before re-allocating
                self._ws = torch.empty(n, dtype=torch.uint8, device=self.device)
            self._ws_key = key
        return self._ws

    def _flags(self, general_tail=False, share_cu=0, block_tail=False):
        return ((0 if self.use_side_streams else L.FWD_SERIAL) | (L.FWD_GENERAL_TAIL if general_tail else 0) | (L.FWD_BLOCK_TAIL if (block_tail or self.block_tail) else 0)
                | (L.FWD_SHARE_CU_CONV4 if share_cu == 1 else L.FWD_YIELD_CU_CONV4 if share_cu == 2 else
                   (L.FWD_YIELD_CU_CONV4 | L.FWD_LATE_DENSE1) if share_cu == 3 else 0))

    def forward(self, x, out=None, want_taps=False, profile=False, general_tail=False, block_tail=False):
        """x: float32 NCHW [B,3,H,W] on this engine's GPU -> float32 NCHW [B,3,H,W].  Asynchronous on the current stream: enqueues the
        launches and returns; never times or synchronises anything (tune() is a separate, explicit call).
        block_tail / general_tail: decoder.final_dense as ONE launch (csrc/final_block.hip) / as the general five-launch chain instead of the
        four-launch chain with the transition folded in (bit-identical / within the order of one fp32 sum): A/B runs and tests."""
        if self.params is None:
            raise L.MdieError("CdanEngine.forward before load(state_dict)")
        _require_gpu(x, "CdanEngine.forward")
        if x.dim() != 4 or x.shape[1] != 3:
            raise L.MdieError(f"expected input [B,3,H,W], got {tuple(x.shape)}")
        x = x.to(torch.float32).contiguous()
        B, _, H, W = x.shape
        ws = self._workspace(B, H, W)
        share = self.form(B, H, W)
        if out is None and not want_taps and not profile:
            # the plain path goes through the registered operator (torch.ops.mdie.cdan_forward, ops.py)
            from . import ops  # noqa: F401  (registers the library)
            aux = self._aux.value if (self.use_side_streams and self._aux) else 0
            return torch.ops.mdie.cdan_forward(x, self.params, ws, self.dtype, aux or 0, self._flags(general_tail, share, block_tail))
        y = out if out is not None else torch.empty_like(x)
        d = L.CdanFwdDesc()
        d.dtype, d.B, d.H, d.W = self.dtype, B, H, W
        d.params, d.x, d.y = self.params.data_ptr(), x.data_ptr(), y.data_ptr()
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
        d.flags = self._flags(general_tail, share, block_tail)
        d.aux = self._aux if self.use_side_streams else None
        taps = (L.Tap * len(L.TAP_NAMES))() if want_taps else None
        if taps is not None:
            d.taps = taps
        if profile:
            cap = 256
            ms, kind, n, info = (C.c_float * cap)(), (C.c_int * cap)(), C.c_int(0), (L.LaunchInfo * cap)()
            d.launch_ms, d.launch_kind, d.max_launches, d.n_launches, d.launch_info = ms, kind, cap, C.pointer(n), info
        with torch.cuda.device(self.device):
            L.check(L.lib.mdie_cdan_forward(C.byref(d), _stream_ptr(self.device)), "mdie_cdan_forward")
        extras = {}
        if taps is not None:
            extras["taps"] = {name: self._read_tap(taps[i], B) for i, name in enumerate(L.TAP_NAMES) if taps[i].ptr}
        if profile:
            extras["launches"] = [(L.KERNEL_KINDS[kind[i]], float(ms[i])) for i in range(n.value)]
            # per launch: the layer it belongs to and its share of the SURVEY 8d model (bytes, FLOPs) -- from the engine's own launch list
            extras["launch_info"] = [(info[i].label.decode(), float(info[i].alg_bytes), float(info[i].flops)) for i in range(n.value)]
        return (y, extras) if extras else y

    def _read_tap(self, tap, B):
        out = torch.empty(B, tap.channels, tap.H, tap.W, dtype=torch.float32, device=self.device)
        L.check(L.lib.mdie_nhwc_to_nchw(self.dtype, B, tap.channels, tap.H, tap.W, C.c_void_p(tap.ptr), out.data_ptr(),
                                        _stream_ptr(self.device)), "mdie_nhwc_to_nchw")
        return out


def routed_shard(labels, rank, world, tasks):
    """which images of a routed batch a rank runs: tasks are dealt to ranks by their position in the sorted task list (rank r owns
    tasks r, r + world, ...), an image follows its task, un-routed images (label None) are dealt round-robin.  Every rank holds
    only its own tasks' weight sets; no collective is involved ("group images by routed task per rank")."""
    order = {t: i for i, t in enumerate(sorted(tasks, key=str))}
    mine, k = [], 0
    for i, t in enumerate(labels):
        if t is None:
            if k % world == rank:
                mine.append(i)
            k += 1
        elif order[t] % world == rank:
            mine.append(i)
    return mine


def routed_slice(n, rank, world):
    """which images of a routed batch of n a rank runs in chain mode: every rank holds all the weight sets, so the batch is cut into
    `world` contiguous slices whose sizes differ by at most one -- whatever the labels; no collective is involved."""
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return list(range(lo, lo + q + (1 if rank < r else 0)))


class RoutedEngine:
    """Classifier-routed inference (BASELINE configs[3]): one weight set per degradation task
    (the reference trains one CDAN per config/*.json), every image labelled with its task by a router
    (mdie_amd/router.py; classification/train_multilabel_classifier.py:251-253 for the thresholds).

    ONE launch chain over the whole batch.  The task weight sets are rows of one device table; the batch is ordered by task and every
    kernel of the chain looks up its image's row through `mdie_cdan_fwd_desc.blob_delta` (a device array of byte offsets, one per image)
    -- 39 launches per batch whatever the grouping, the side branches on the engine's aux streams as in CdanEngine.  Bit-identical to
    per-task engines (tests/test_gpu_parity.py::test_routed_*): a kernel is chosen by (layer, map) alone and the several-weight-sets
    variants keep the arithmetic order of the plain ones.
    (Rounds 3-4 also had a "groups" mode -- one 39-launch chain per task group on its own stream, enqueued by one host thread each:
    16 k against 27 k images/s, and its per-group hipGraph variant took the process down once (profiles/r04e_graph_replay_segfault.log).
    Removed in round 5: the chain serves every grouping, from ONE host thread.)"""

    def __init__(self, device, precision="bf16"):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise L.MdieError(f"RoutedEngine needs a GPU device, got {self.device} (no CPU fallback)")
        self.dtype = dtype_id(precision)
        self.blobs = {}
        self._table = None           # (rows [ntask, stride] uint8, {task: row})
        self._ws = None
        self._aux = C.c_void_p(0)
        with torch.cuda.device(self.device):
            L.check(L.lib.mdie_aux_create(C.byref(self._aux)), "mdie_aux_create")

    def __del__(self):
        try:
            if self._aux:
                L.lib.mdie_aux_destroy(self._aux)
                self._aux = C.c_void_p(0)
        except Exception:
            pass

    def load_task(self, task, state_dict):
        self.blobs[task] = pack_checkpoint(state_dict, self.dtype).to(self.device)
        self._table = None
        return self

    def _weight_table(self):
        """the task weight sets as rows of one device buffer (row stride a multiple of 256 bytes): an image's byte offset
        from row 0 is its task's row x stride"""
        if self._table is None:
            tasks = sorted(self.blobs, key=str)
            n = max(b.numel() for b in self.blobs.values())
            if any(b.numel() != n for b in self.blobs.values()):
                raise L.MdieError("RoutedEngine: the task weight sets differ in size (one architecture per engine)")
            stride = (n + 255) // 256 * 256
            rows = torch.zeros(len(tasks), stride, dtype=torch.uint8, device=self.device)
            for i, t in enumerate(tasks):
                rows[i, :n].copy_(self.blobs[t].view(torch.uint8).reshape(-1))
            self._table = (rows, {t: i for i, t in enumerate(tasks)}, stride)
        return self._table

    def _chain(self, x, labels, routed):
        """one launch chain: images ordered by task, delta[i] = row(task of image i) x stride"""
        B, _, H, W = x.shape
        rows, row_of, stride = self._weight_table()
        order = sorted(routed, key=lambda i: row_of[labels[i]])
        n = len(order)
        nbytes = L.lib.mdie_cdan_workspace_bytes(self.dtype, n, H, W)
        if nbytes == 0:
            raise L.MdieError(f"unsupported input extent {n}x3x{H}x{W}: H and W must be multiples of 8")
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = None
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        # one host -> device copy carries both the order and the offsets
        meta = torch.tensor([order, [row_of[labels[i]] * stride for i in order]], dtype=torch.int64).to(self.device, non_blocking=True)
        identity = n == B and order == list(range(B))
        xs = x if identity else x.index_select(0, meta[0])
        ys = torch.empty_like(xs)
        d = L.CdanFwdDesc()
        d.dtype, d.B, d.H, d.W = self.dtype, n, H, W
        d.params, d.x, d.y = rows.data_ptr(), xs.data_ptr(), ys.data_ptr()
        d.workspace, d.workspace_bytes = self._ws.data_ptr(), self._ws.numel()
        d.flags, d.aux = (L.FWD_YIELD_CU_CONV4 if _share_cu_eligible(self.dtype, n, H, W) else 0), self._aux      # (untimed default: CdanEngine.tune)
        d.blob_delta = meta[1].data_ptr()
        L.check(L.lib.mdie_cdan_forward(C.byref(d), _stream_ptr(self.device)), "mdie_cdan_forward")
        self._keep = (meta, xs)          # (alive until the next call: the chain reads them asynchronously)
        if identity:
            return ys
        if n == B:
            out = torch.empty_like(x)
        else:
            out = x.clone()              # the router found no degradation: those images pass through unchanged
        out.index_copy_(0, meta[0], ys)
        return out

    def forward(self, x, labels):
        """x: float32 NCHW [B,3,H,W] on the GPU; labels: B task keys (host side; None = pass through).  Returns [B,3,H,W] in input order."""
        _require_gpu(x, "RoutedEngine.forward")
        labels = list(labels)
        if x.dim() != 4 or x.shape[1] != 3 or len(labels) != x.shape[0]:
            raise L.MdieError(f"RoutedEngine.forward: input {tuple(x.shape)} with {len(labels)} labels")
        missing = sorted({str(t) for t in labels if t is not None and t not in self.blobs})
        if missing:
            raise L.MdieError(f"RoutedEngine.forward: no weights loaded for task(s) {missing}")
        B, _, H, W = x.shape
        x = x.to(torch.float32).contiguous()
        routed = [i for i, t in enumerate(labels) if t is not None]
        with torch.cuda.device(self.device):
            return self._chain(x, labels, routed) if routed else x.clone()


# ---- thin per-op wrappers (used by the parity tests; same entry points the plan calls) -----------------------------
def to_nhwc(x, dtype):
    _require_gpu(x, "to_nhwc")
    B, Cc, H, W = x.shape
    out = torch.empty(B, H, W, Cc, dtype=TORCH_DTYPE[dtype], device=x.device)
    L.check(L.lib.mdie_nchw_to_nhwc(dtype, B, Cc, H, W, x.contiguous().data_ptr(), out.data_ptr(), _stream_ptr(x.device)),
            "mdie_nchw_to_nhwc")
    return out


def to_nchw(x_nhwc, dtype):
    B, H, W, Cc = x_nhwc.shape
    out = torch.empty(B, Cc, H, W, dtype=torch.float32, device=x_nhwc.device)
    L.check(L.lib.mdie_nhwc_to_nchw(dtype, B, Cc, H, W, x_nhwc.data_ptr(), out.data_ptr(), _stream_ptr(x_nhwc.device)),
            "mdie_nhwc_to_nchw")
    return out


def pack_conv_weight(w, dtype, transposed=False, cout_stored=None, cin_stored=None, split=None, gap=0):
    w = np.ascontiguousarray(w.detach().cpu().float().numpy())
    ks = w.shape[2]
    cout, cin = (w.shape[1], w.shape[0]) if transposed else (w.shape[0], w.shape[1])
    cout_stored = cout_stored or (cout + 15) // 16 * 16
    cin_stored = cin_stored or (cin + 15) // 16 * 16
    split = cin if split is None else split
    n = L.lib.mdie_conv_weight_bytes(dtype, ks, cin_stored, cout_stored)
    dst = torch.zeros(n, dtype=torch.uint8)
    L.check(L.lib.mdie_pack_conv_weight(dtype, ks, int(transposed), w.ctypes.data, cout, cin, cout_stored, cin_stored,
                                        split, gap, dst.data_ptr()), "mdie_pack_conv_weight")
    return dst


def conv_fwd(segments, weight_packed, post_scale, post_shift, *, dtype, ksize, cout, act=L.ACT_NONE, pool=False,
             pre_scale=None, pre_shift=None, residual=None, out=None, out_view=None):
    """segments: list of NHWC tensors [B,H,W,Ci] (Ci % 16 == 0).  Returns NHWC [B,Ho,Wo,cout]."""
    x0 = segments[0]
    B, H, W, _ = x0.shape
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    if out is None:
        out = torch.empty(B, Ho, Wo, cout, dtype=TORCH_DTYPE[dtype], device=x0.device)
    d = L.ConvDesc()
    d.dtype, d.B, d.H, d.W, d.ksize = dtype, B, H, W, ksize
    d.nseg = len(segments)
    cin = 0
    for i, s in enumerate(segments):
        d.inp[i] = L.Seg(s.data_ptr(), s.shape[3], s.stride(2))
        cin += s.shape[3]
    d.cin, d.cout = cin, cout
    d.pre_scale, d.pre_shift = _ptr(pre_scale), _ptr(pre_shift)
    d.weight, d.post_scale, d.post_shift = _ptr(weight_packed), _ptr(post_scale), _ptr(post_shift)
    d.act, d.pool = act, int(pool)
    d.residual = _ptr(residual)
    d.res_stride = residual.stride(2) if residual is not None else 0
    target = out_view if out_view is not None else out
    d.out, d.out_stride = target.data_ptr(), target.stride(2)
    d.out_nchw3 = None
    L.check(L.lib.mdie_conv_fwd(C.byref(d), _stream_ptr(x0.device)), "mdie_conv_fwd")
    return out


def conv_first_fwd(x_nchw, w, post_scale, post_shift, *, dtype, act=L.ACT_NONE, pool=False):
    """x_nchw: fp32 [B,3,H,W]; w: nn.Conv2d weight [cout,3,3,3].  Returns NHWC [B,Ho,Wo,cout_stored]."""
    _require_gpu(x_nchw, "conv_first_fwd")
    x_nchw = x_nchw.contiguous()
    B, _, H, W = x_nchw.shape
    cout = w.shape[0]
    cst = (cout + 15) // 16 * 16
    wn = np.ascontiguousarray(w.detach().cpu().float().numpy())
    packed = torch.zeros(L.lib.mdie_conv_first_weight_bytes(dtype, cst), dtype=torch.uint8)
    L.check(L.lib.mdie_pack_conv_first_weight(dtype, wn.ctypes.data, cout, cst, packed.data_ptr()), "mdie_pack_conv_first_weight")
    packed = packed.to(x_nchw.device)
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    out = torch.empty(B, Ho, Wo, cst, dtype=TORCH_DTYPE[dtype], device=x_nchw.device)
    d = L.ConvFirstDesc()
    d.dtype, d.B, d.H, d.W, d.x = dtype, B, H, W, x_nchw.data_ptr()
    d.weight, d.post_scale, d.post_shift = packed.data_ptr(), post_scale.data_ptr(), post_shift.data_ptr()
    d.cout, d.act, d.pool = cst, act, int(pool)
    d.out, d.out_stride = out.data_ptr(), cst
    L.check(L.lib.mdie_conv_first_fwd(C.byref(d), _stream_ptr(x_nchw.device)), "mdie_conv_first_fwd")
    return out


def cbam_fwd(x, w1, b1, w2, b2, w7, bn, *, dtype, mul=None, channel_only=False):
    B, H, W, Cc = x.shape
    out = torch.empty_like(x)
    n = L.lib.mdie_cbam_workspace_bytes(B, H, W, Cc)
    ws = torch.empty(n, dtype=torch.uint8, device=x.device)
    d = L.CbamDesc()
    d.dtype, d.B, d.H, d.W, d.C = dtype, B, H, W, Cc
    d.x, d.x_stride = x.data_ptr(), x.stride(2)
    d.w1, d.b1, d.w2, d.b2, d.w7, d.bn = (t.data_ptr() for t in (w1, b1, w2, b2, w7, bn))
    d.mul = _ptr(mul)
    d.mul_stride = mul.stride(2) if mul is not None else 0
    d.out, d.out_stride = out.data_ptr(), out.stride(2)
    d.workspace, d.workspace_bytes = ws.data_ptr(), n
    d.pool_partial, d.pool_slabs = None, 0
    fn = L.lib.mdie_cbam_channel_only_fwd if channel_only else L.lib.mdie_cbam_fwd
    L.check(fn(C.byref(d), _stream_ptr(x.device)), "mdie_cbam_fwd")
    return out


def upsample2x_add(lo, skip, *, dtype):
    B, H, W, Cc = lo.shape
    out = torch.empty_like(skip)
    L.check(L.lib.mdie_upsample2x_add(dtype, B, H, W, Cc, lo.data_ptr(), lo.stride(2), skip.data_ptr(), skip.stride(2),
                                      out.data_ptr(), out.stride(2), _stream_ptr(lo.device)), "mdie_upsample2x_add")
    return out


def _tensor_array(state_dict):
    keep, entries = [], []
    for name, value in state_dict.items():
        if not torch.is_tensor(value) or not value.dtype.is_floating_point:
            continue
        a = np.ascontiguousarray(value.detach().to("cpu", torch.float32).numpy())
        keep.append(a)
        entries.append(L.Tensor(name.encode(), a.ctypes.data, a.size))
    return (L.Tensor * len(entries))(*entries), len(entries), keep
