"""Training-mode CDAN (`Model.train_step`, models/model.py:138-227).

Every tensor-sized operation of the network trunk runs in libmdie_hip.so; torch.autograd is only the tape
that orders three block-level Functions:

  _ConvBnFn   encoder ConvBlock: conv3x3 -> batch-stat BN -> ReLU -> maxpool -> dropout   (models/cdan.py:15-19,74-79)
  _DenseFn    DenseBlock: 4 x [BN -> ReLU -> conv3x3] -> BN -> ReLU -> conv1x1 [-> sigmoid] (models/cdan.py:32-53,155-157)
  _DeconvFn   decoder stage: ConvTranspose3x3 -> BN -> ReLU [-> bilinear x2] + skip       (models/cdan.py:127-154)
  _CbamFn     CBAM (channel gate, spatial gate with batch-stat BN) [* dense_k]            (models/cbam.py:37-95)

each of which calls the HIP convolution (forward / dgrad / wgrad, csrc/conv.hip, csrc/train.hip) and the fused
BatchNorm kernels of csrc/bn.hip in both directions.  The pre-activation BN + ReLU of the dense layers is never
materialised: it is the staging prologue of the convolution and of its weight-gradient kernel, and a segment's
batch statistics are computed once, not once per consuming layer.  A fourth Function, _CbamFn, is CBAM (with the
`out *= dense_k` that follows it) on csrc/cbam_train.hip.  Nothing here touches the CPU or the oracle, and no
tensor-sized PyTorch operator is left on the path.  Tensors are `channels_last`, which IS the engine's NHWC layout.

Reference semantics kept: BatchNorm uses batch statistics and updates running statistics with momentum 0.1
(0.01 in CBAM's spatial gate, models/cbam.py:11), eps 1e-5; dropout p=0.2 after each encoder stage with the
dense blocks fed from the PRE-dropout tensor (models/cdan.py:76-79); ConvTranspose2d weights stay in their
[Cin, Cout, kh, kw] checkpoint layout.  A convolution bias that feeds a batch-statistic BatchNorm has an exactly
zero gradient (the mean subtraction removes it): those gradients are returned as exact zeros.
"""
import ctypes as C

import torch
import torch.nn.functional as F

from . import engine as E
from . import lib as L

EPS = 1e-5
# fp32 running sums for the DenseBlock segment gradients under 16-bit storage (MDIE_TRAIN_ACC32=1).  OFF by default: measured on
# MI355X (whole network, 2x3x64x64, vs the fp64 oracle) the gradient direction does not move -- bf16 median cosine 0.98699 with,
# 0.98708 without, worst 0.897 / 0.893 (the bottleneck gate's arg-max routing) -- because what separates bf16 from the oracle is
# the rounding of the FORWARD activations (fp16, 8x finer, gives 0.9987 / 0.993 with the same five bf16-style roundings), while
# the fp32 traffic costs 6 % of a 512x512 step (11.57 vs 10.88 ms, B = 8).
ACC32 = __import__("os").environ.get("MDIE_TRAIN_ACC32", "0") == "1"
BN_REDUCE_IN_DGRAD = __import__("os").environ.get("MDIE_TRAIN_BNRED", "1") == "1"   # the DenseBlock layers' BatchNorm-backward sums from the input-gradient convolution's epilogue


def _cl(t):
    return t.contiguous(memory_format=torch.channels_last)


def _nhwc(t):
    """[B,C,H,W] tensor whose memory is NHWC (possibly a channel slice of one) -> (data_ptr, C, pixel stride in elements)."""
    B, Cc, H, W = t.shape
    ps = t.stride(3) if W > 1 else (t.stride(2) if H > 1 else (t.stride(0) if B > 1 else Cc))
    assert (Cc == 1 or t.stride(1) == 1) and (H == 1 or W == 1 or t.stride(2) == W * ps) and (B == 1 or t.stride(0) == H * W * ps), \
        "engine tensors are NHWC"
    return t.data_ptr(), Cc, ps


def _sp(dev):
    return E._stream_ptr(dev)


def _f32(t):
    t = t.detach()
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()


# Every uninitialised allocation of this file goes through _raw / _raw_like.  MDIE_TRAIN_POISON=1 (a debug switch, also settable as
# train.POISON) fills each one before use -- NaN for floating point, 0xFF bytes for the uint8 workspaces (NaN when read as fp32 / bf16 /
# fp16 partials, -1 as a count), 0x7F7F7F7F for integers -- so that a kernel that READS memory no launch of the step has written turns
# the step's losses / gradients into NaN deterministically, whatever the allocator happens to hand back (round-3 finding 6: gradients
# that depended on the schedule; allocator-address dependence is what the earlier poison -- two CBAM arenas only -- did not cover).
POISON = __import__("os").environ.get("MDIE_TRAIN_POISON", "0") == "1"


def _poison(t):
    if t.numel():
        if t.is_floating_point():
            t.fill_(float("nan"))
        elif t.dtype == torch.uint8:
            t.fill_(0xFF)
        else:
            t.fill_(0x7F7F7F7F if t.dtype in (torch.int32, torch.int64) else 0x7F)
    return t


def _raw(*shape, **kw):
    t = torch.empty(*shape, **kw)
    return _poison(t) if POISON else t


def _raw_like(x, **kw):
    t = torch.empty_like(x, **kw)
    return _poison(t) if POISON else t


def _empty(dt, B, Cc, H, W, dev):
    return _raw(B, Cc, H, W, dtype=E.TORCH_DTYPE[dt], device=dev, memory_format=torch.channels_last)


class _PackPlan:
    """Every weight repack of a training step in ONE launch.

    A step repacks each convolution's fp32 parameter twice -- the forward form and the flipped / transposed input-gradient
    form -- into the engine's K-chunk layout, because the optimizer has just changed it: 55 launches of 4-5 us on a GPU-bound
    step.  The plan learns the jobs from the first (eager) step, whose `_pack` calls run one by one and register themselves;
    from then on `forward_train` starts with `run()` -- one mdie_pack_conv_weights_batch launch over the device-resident job table
    -- and `_pack` hands out the buffers that launch has filled.  A job is keyed by the SOURCE tensor's address, so only
    weights that live where they lived (the module's parameters) are served from the plan; anything else -- the zero-padded
    copies decoder.final_dense builds per step -- takes the single-launch path, as does any job first seen after the table was
    uploaded (it joins the table at the next step)."""

    def __init__(self, dt):
        self.dt, self.jobs, self.table, self.n_uploaded, self.filled, self.param_ptrs = dt, {}, None, 0, False, frozenset()
        self.retired = []     # superseded job tables: a CapturedStep that recorded the batch launch replays with the OLD table's address

    def lookup(self, key):
        j = self.jobs.get(key)
        return j[1] if (j is not None and self.filled and j[2] < self.n_uploaded) else None

    def register(self, key, src, dst, args):
        if key not in self.jobs:
            self.jobs[key] = (src, dst, len(self.jobs), args)      # (keeps src and dst alive: their addresses are in the table)

    def run(self, dev):
        self.filled = False
        if not self.jobs:
            return
        if self.n_uploaded != len(self.jobs) and not torch.cuda.is_current_stream_capturing():
            arr = (L.PackJob * len(self.jobs))()
            for src, dst, i, args in self.jobs.values():
                arr[i] = L.PackJob(src.data_ptr(), dst.data_ptr(), *args)
            if self.table is not None:
                self.retired.append(self.table)       # (a few hundred bytes each; freed with the plan) -- its jobs stay valid: jobs are only ever appended
            self.table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
            self.n_uploaded = len(self.jobs)
        if self.n_uploaded:
            L.check(L.lib.mdie_pack_conv_weights_batch(self.dt, self.table.data_ptr(), self.n_uploaded, _sp(dev)), "mdie_pack_conv_weights_batch")
            self.filled = True


_PLAN = None     # the plan of the network whose step is running (forward_train sets it; backward runs inside the same step)

# The bias of a convolution that feeds a BatchNorm has an exactly-zero gradient (the batch mean absorbs it): 24 of the 35
# convolutions.  Autograd still wants a tensor per parameter; they are slices of ONE zero-filled arena per step (one fill
# launch instead of 24 -- each slice belongs to one parameter, so in-place users of .grad -- GradScaler.unscale_, gradient
# clipping -- see ordinary separate tensors).
_ZERO_ARENA = None   # [tensor, next free element]


def _new_zero_arena(dev, n=4096):
    global _ZERO_ARENA
    _ZERO_ARENA = [torch.zeros(n, dtype=torch.float32, device=dev), 0]


# Data parallel: where a gradient is WRITTEN.  With a GradBuckets active (`_GRAD_SINK`), every kernel that produces a parameter
# gradient -- the weight-gradient reduce, the BatchNorm backward's dgamma / dbeta, the CBAM backward, the exactly-zero biases --
# writes straight into that parameter's slice of its flat all-reduce bucket, and the Function returns a fresh VIEW of the slice:
# autograd's AccumulateGrad takes it over as `.grad` (one reference, the parameter's layout), so `.grad` IS the bucket memory
# and the exchange moves no gradient in or out (round 3: 140 copies in + 140 back per step).  Only while `param.grad is None`
# (zero_grad(set_to_none=True), what Model.train_step does): an existing .grad is accumulated into by autograd as usual.  And only
# ONCE per parameter per backward (GradBuckets.claim, keyed by the autograd graph task): when one backward produces two gradients of
# one parameter (the network applied twice in one graph, weight sharing), `.grad` is still None at the second producer while the first
# gradient -- the slice -- sits un-summed in autograd's input buffer; a second kernel writing the same slice would overwrite it and
# autograd would add the slice to itself (2 x the second contribution).  The second sighting therefore gets a fresh tensor, autograd sums
# the two, and the hook's copy path puts the sum into the slice.  Outside an engine-run backward (no graph task) nothing is claimed.
# (`torch.autograd.grad` with a sink active returns views of bucket memory for sink-aware parameters: valid until the next backward.)
_GRAD_SINK = None


def _gout(param, shape, dev):
    """the tensor a kernel writes `param`'s gradient into: the parameter's bucket slice (as a fresh view) or a new tensor"""
    sink = _GRAD_SINK
    if sink is not None and param is not None and param.grad is None and sink.claim(param):
        v = sink.view_of(param)
        if v is not None and tuple(v.shape) == tuple(shape):
            return v
    return _raw(shape, dtype=torch.float32, device=dev)


def _zero_grad_vec(n, dev, param=None):
    sink = _GRAD_SINK
    if sink is not None and param is not None and param.grad is None and sink.claim(param):
        v = sink.zero_view_of(param)          # (a slice no kernel ever writes: zero since the bucket was made, all-reduced zeros stay zero)
        if v is not None and v.numel() == n:
            return v
    a = _ZERO_ARENA
    if a is None or a[0].device != dev or a[1] + n > a[0].numel():
        return torch.zeros(n, dtype=torch.float32, device=dev)
    a[1] += n
    return a[0][a[1] - n:a[1]]


def _pack(dt, w32, ks, transposed, cout, cin, cout_st=None, cin_st=None, split=None, gap=0, out_split=None, out_gap=0):
    """the engine's K-chunk form of an fp32 weight (mdie_pack_job: `split` / `gap` place the real input channels inside a wider
    stored input, `out_split` / `out_gap` the real output channels inside a wider stored output)"""
    cout_st, cin_st = cout_st or cout, cin_st or cin
    args = (ks, int(bool(transposed)), cout, cin, cout_st, cin_st, cin if split is None else split, gap, cout if out_split is None else out_split, out_gap)
    key = (w32.data_ptr(),) + args
    plan = _PLAN if (_PLAN is not None and _PLAN.dt == dt) else None
    if plan is not None:
        hit = plan.lookup(key)
        if hit is not None:
            return hit
    n = L.lib.mdie_conv_weight_bytes(dt, ks, cin_st, cout_st)
    dst = _raw(n, dtype=torch.uint8, device=w32.device)
    job = L.PackJob(w32.data_ptr(), dst.data_ptr(), *args)
    L.check(L.lib.mdie_pack_conv_weight_job(dt, C.byref(job), _sp(w32.device)), "mdie_pack_conv_weight_job")
    if plan is not None and w32.data_ptr() in plan.param_ptrs:
        plan.register(key, w32, _raw_like(dst), args)
    return dst


_ONES = {}


def _ones(n, dev):
    key = (n, dev)
    if key not in _ONES:
        _ONES[key] = torch.ones(n, dtype=torch.float32, device=dev)
    return _ONES[key]


_ZEROS = {}


def _zeros(n, dev):
    """read-only zero vector (bias of the gradient convolutions)"""
    key = (n, dev)
    if key not in _ZEROS:
        _ZEROS[key] = torch.zeros(n, dtype=torch.float32, device=dev)
    return _ZEROS[key]


# How the training step's wide (conv_wide) layers treat their CUs: 0 = one persistent workgroup per CU until the launch ends (the default: fastest
# alone), 2 = two shorter runs per CU -- every CU returns to the dispatcher half way -- 1 = conv_kernel (mdie_conv_desc.share_cu; the forms are
# bit-identical).  On ONE GPU nothing runs beside these layers and 0 wins; under data parallelism RCCL's all-reduce kernels need CUs while backward is
# still running, and a persistent workgroup that holds its CU until its launch ends is exactly what starved the DenseBlock branches in inference
# (DESIGN section 4 / 7).  MDIE_TRAIN_SHARE_CU=2 is the switch for the first multi-GPU run to A/B against exposed all-reduce time; unmeasured here.
WIDE_SHARE_CU = int(__import__("os").environ.get("MDIE_TRAIN_SHARE_CU", "0"))


def _conv_raw(dt, segs, packed, bias_st, ks, cout_st, out, pre=None, act=L.ACT_NONE, out_nchw3=None, planar=False, bnred=None):
    """out = act(conv_k(relu(cat(segs) * pre_scale + pre_shift)?) + bias); out: NHWC view with >= cout_st channels, or
    (planar) a [cout_st / 16, B*H*W, 16] tensor: one plane per 16 output channels (mdie_conv_desc.out_group_stride)."""
    B, _, H, W = segs[0].shape
    d = L.ConvDesc()
    d.dtype, d.B, d.H, d.W, d.ksize = dt, B, H, W, ks
    d.nseg = len(segs)
    cin = 0
    for i, s in enumerate(segs):
        ptr, c, st = _nhwc(s)
        d.inp[i] = L.Seg(ptr, c, st)
        cin += c
    d.cin, d.cout = cin, cout_st
    d.pre_scale, d.pre_shift = (pre[0].data_ptr(), pre[1].data_ptr()) if pre is not None else (None, None)
    d.weight, d.post_scale, d.post_shift = packed.data_ptr(), _ones(cout_st, out.device).data_ptr(), bias_st.data_ptr()
    d.act, d.pool = act, 0
    d.residual, d.res_stride = None, 0
    if planar:
        d.out, d.out_stride, d.out_group_stride = out.data_ptr(), 16, out.stride(0)
    else:
        d.out, d.out_stride = out.data_ptr(), out.stride(3)
    d.out_nchw3 = out_nchw3.data_ptr() if out_nchw3 is not None else None
    d.share_cu = WIDE_SHARE_CU
    if bnred is not None:     # (x segments, scale, shift, partial): the BatchNorm-ReLU backward sums of x ride on this input-gradient convolution
        xsegs, bsc, bsh, partial = bnred
        r = L.BnReduceFuse()
        r.nseg = len(xsegs)
        for i, s in enumerate(xsegs):
            ptr, c, st = _nhwc(s)
            r.x[i] = L.Seg(ptr, c, st)
        r.scale, r.shift, r.partial, r.partial_bytes = bsc.data_ptr(), bsh.data_ptr(), partial.data_ptr(), partial.numel() * 4
        d.bnred = C.pointer(r)
    L.check(L.lib.mdie_conv_fwd(C.byref(d), _sp(out.device)), "mdie_conv_fwd")


def _pad_vec(v, n):
    v = _f32(v)
    return v if v.numel() == n else F.pad(v, (0, n - v.numel()))


# Weight gradients on a stream of their own -- OPT-IN (MDIE_TRAIN_WGRAD_STREAM=1), see the last paragraph.  Nothing in backward
# waits for a dW -- the chain is input gradient -> BatchNorm backward -> input gradient -- and that chain is full of kernels that
# leave the GPU nearly idle (the *_final folds, the gates, the 32x32 / 64x64 layers): the weight-gradient kernels (1.8 ms of an
# 8.4 ms step) run under them.  Fork: the side stream waits for the main stream at the point of the call (dy and every input are
# complete in main-stream order); join: ONE wait of the main stream for the side stream when backward ends (an autograd-engine
# callback queued from the first call of each backward), so the optimizer, the GradScaler, a gradient exchange that runs after
# backward -- everything that reads .grad -- comes after.
# Not taken when something reads or combines the gradient DURING backward: a parameter that already holds a gradient (accumulation
# adds in place on the main stream), a parameter with a tensor / post-accumulate hook of its own, GradBuckets' per-parameter hooks,
# a parameter that has ALREADY received a side-stream dW in this backward (the network applied twice in one graph, weight sharing:
# autograd sums the two on the main stream), or a backward that records a graph itself (create_graph=True: AccumulateGrad clones
# instead of stealing).  And only where it pays (measured, bf16, B = 8, same box): eager 512x512 8.69 -> 8.11 ms; as a hipGraph
# 8.71 -> 8.73 and 256x256 3.85 -> 4.21 as a graph, 7.06 -> 8.87 eager (host-bound) -- so: eager steps of at least
# WGRAD_STREAM_MIN_PIXELS input pixels, never under capture.
# WHY IT IS OFF BY DEFAULT (round 4).  In round 3 one box showed, in 4 of 12 rounds of three steps at 8x512x512, the bottleneck
# CBAM's channel-gate MLP gradients (and what is downstream of them) differing from the single-stream schedule.  Those gradients
# are computed entirely on the MAIN stream (cbt_bwd3 -> cbt_gate_bwd -> cbt_gate_final); the cause was not found: profiles/LEDGER.md
# (rounds 1-4 section 4, finding 6; round 5) lists what reading the kernels' ISA, the host-side fork, a hardware probe
# (tools/pkfma_probe.hip) and a NaN poison of every uninitialised allocation of the step (MDIE_TRAIN_POISON, round 5) exclude.
# The default training schedule is the single stream; this switch is EXPERIMENTAL.
WGRAD_STREAM = __import__("os").environ.get("MDIE_TRAIN_WGRAD_STREAM", "0") == "1"
WGRAD_STREAM_MIN_PIXELS = 8 * 384 * 384
_wgrad_side_this_step = False  # forward_train decides per step
_WGRAD_SIDE = {}
_wgrad_hooks_active = 0        # GradBuckets with hooks registered
_wgrad_join_task = {}          # device -> id of the autograd graph task (= one backward) whose join callback is queued
_wgrad_seen = {}               # device -> (graph task id, ids of the parameters that have received a side-stream dW in it)


def _wgrad_side_stream(dev):
    st = _WGRAD_SIDE.get(dev)
    if st is None:
        st = _WGRAD_SIDE[dev] = torch.cuda.Stream(dev)
    return st


def _graph_task():
    return torch._C._current_graph_task_id() if hasattr(torch._C, "_current_graph_task_id") else None


def _first_sighting(dev, param):
    """True the first time `param` asks for a side-stream dW inside the running backward (one autograd graph task)."""
    task = _graph_task()
    if task is None or task < 0:
        return False               # not inside an engine-run backward (or a torch without the query): stay on the main stream
    seen = _wgrad_seen.get(dev)
    if seen is None or seen[0] != task:
        seen = _wgrad_seen[dev] = (task, set())
    if id(param) in seen[1]:
        return False
    seen[1].add(id(param))
    return True


def _queue_wgrad_join(dev, main):
    """`main`: the stream backward runs on, taken inside a Function.backward (the engine has set it to the forward's stream there).
    The callback itself may run on an autograd worker thread whose current stream is the device's default one -- under graph
    capture that is NOT the capturing stream -- so the stream to join is fixed here, not looked up there."""
    task = _graph_task()
    if task is not None and task >= 0 and _wgrad_join_task.get(dev) == task:
        return                       # this backward already has its join (keyed by the task, so a backward that died half way leaves nothing behind)
    _wgrad_join_task[dev] = task

    def join():
        main.wait_stream(_wgrad_side_stream(dev))

    torch.autograd.Variable._execution_engine.queue_callback(join)


def join_weight_gradients(dev):
    """make the current stream wait for weight gradients still running on the side stream (idempotent; backward's own callback
    has normally done it already)"""
    st = _WGRAD_SIDE.get(torch.device(dev) if not isinstance(dev, torch.device) else dev)
    if st is not None and not torch.cuda.is_current_stream_capturing():    # (nothing is ever forked under capture; a capturing stream
        torch.cuda.current_stream(dev).wait_stream(st)                      #  must not wait on an event of a stream outside the capture)


def _wgrad(dt, segs, dy, w_shape, ks, transposed, cin, cout, cout_st, pre=None, split=None, gap=0, param=None):
    """dW in the parameter's layout from the convolution's input segments and dy (stored cout_st channels)."""
    dev = dy.device
    if (_wgrad_side_this_step and WGRAD_STREAM and _wgrad_hooks_active == 0 and _GRAD_SINK is None     # (experimental schedule: never together with a gradient exchange)
            and param is not None and param.grad is None
            and not torch.is_grad_enabled()                                                                                   # create_graph: the gradient is cloned / differentiated on the main stream
            and not getattr(param, "_backward_hooks", None) and not getattr(param, "_post_accumulate_grad_hooks", None)     # a hook would read the gradient during backward
            and _first_sighting(dev, param)):                                                                                 # a second dW of one backward is summed with the first on the main stream
        main, side = torch.cuda.current_stream(dev), _wgrad_side_stream(dev)
        side.wait_stream(main)
        _queue_wgrad_join(dev, main)
        with torch.cuda.stream(side):
            dw = _wgrad_launch(dt, segs, dy, w_shape, ks, transposed, cin, cout, cout_st, pre, split, gap, param)
        for t in list(segs) + [dy] + (list(pre) if pre is not None else []):
            t.record_stream(side)          # allocated on the main stream, read on the side stream
        return dw
    if _wgrad_side_this_step and WGRAD_STREAM and param is not None and dev in _WGRAD_SIDE:
        seen = _wgrad_seen.get(dev)
        if seen is not None and seen[0] == _graph_task() and id(param) in seen[1]:
            torch.cuda.current_stream(dev).wait_stream(_WGRAD_SIDE[dev])     # the first dW of this parameter may still be running: autograd adds the two next
    return _wgrad_launch(dt, segs, dy, w_shape, ks, transposed, cin, cout, cout_st, pre, split, gap, param)


def _wgrad_launch(dt, segs, dy, w_shape, ks, transposed, cin, cout, cout_st, pre, split, gap, param=None):
    B, _, H, W = dy.shape
    dev = dy.device
    cin_st = sum(s.shape[1] for s in segs)
    dw = _gout(param, w_shape, dev)
    nws = L.lib.mdie_conv_wgrad_workspace_bytes(B, H, W, ks, cin_st, cout_st)
    ws = _raw(nws, dtype=torch.uint8, device=dev)
    d = L.WgradDesc()
    d.dtype, d.B, d.H, d.W, d.ksize, d.transposed = dt, B, H, W, ks, int(transposed)
    d.nseg = len(segs)
    for i, s in enumerate(segs):
        ptr, c, st = _nhwc(s)
        d.inp[i] = L.Seg(ptr, c, st)
    d.cin, d.cout, d.cout_stored, d.split, d.gap = cin, cout, cout_st, (cin if split is None else split), gap
    ptr, _, st = _nhwc(dy)
    d.dy, d.dy_stride = ptr, st
    d.dw, d.workspace, d.workspace_bytes = dw.data_ptr(), ws.data_ptr(), nws
    d.pre_scale, d.pre_shift = (pre[0].data_ptr(), pre[1].data_ptr()) if pre is not None else (None, None)
    L.check(L.lib.mdie_conv_wgrad(C.byref(d), _sp(dev)), "mdie_conv_wgrad")
    return dw


class _Bn:
    """Batch statistics of one NHWC tensor and the folded constants of one BatchNorm over (a prefix of) them."""

    @staticmethod
    def stats(dt, t, mean, var):
        ptr, c, st = _nhwc(t)
        B, _, H, W = t.shape
        nws = L.lib.mdie_bn_workspace_bytes(c)
        ws = _raw(nws, dtype=torch.uint8, device=t.device)
        L.check(L.lib.mdie_bn_stats(dt, B * H * W, ptr, c, st, mean.data_ptr(), var.data_ptr(), ws.data_ptr(), nws, _sp(t.device)), "mdie_bn_stats")

    @staticmethod
    def stats_fold(dt, t, mv, off, c_st, c_real, split, gap, bn, momentum, partial=None):
        """statistics of t -> mv[:, off : off + C_t], then the fold of `bn` over the stored channels mv[:, :c_st] (which end with
        the new ones) -> consts [3, c_st]: scale, shift, invstd; bn.running_* updated in place.  Two launches
        (mdie_bn_stats_fold), or one when the convolution that wrote t left its partial sums in `partial` (tensor, count)."""
        ptr, c, st = _nhwc(t)
        B, _, H, W = t.shape
        dev = t.device
        k = _raw(3, c_st, dtype=torch.float32, device=dev)
        d = L.BnStatsFoldDesc()
        d.dtype, d.N, d.C, d.stride = dt, B * H * W, c, st
        d.mean, d.var = mv[0, off:].data_ptr(), mv[1, off:].data_ptr()
        if partial is None:
            nws = L.lib.mdie_bn_workspace_bytes(c)
            ws = _raw(nws, dtype=torch.uint8, device=dev)
            d.x, d.workspace, d.workspace_bytes = ptr, ws.data_ptr(), nws
        else:
            d.x, d.workspace, d.workspace_bytes, d.n_partial = None, partial[0].data_ptr(), partial[0].numel() * partial[0].element_size(), partial[1]
        d.C_fold, d.C_real, d.split, d.gap = c_st, c_real, split, gap
        d.fold_mean, d.fold_var, d.gamma, d.beta, d.eps, d.momentum = mv[0].data_ptr(), mv[1].data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr(), EPS, momentum
        d.running_mean, d.running_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
        d.scale, d.shift, d.invstd = k[0].data_ptr(), k[1].data_ptr(), k[2].data_ptr()
        L.check(L.lib.mdie_bn_stats_fold(C.byref(d), _sp(dev)), "mdie_bn_stats_fold")
        return k

    @staticmethod
    def fold(c_st, c_real, split, gap, mean, var, bn, momentum, count, dev):
        """-> consts [3, c_st]: scale, shift, invstd; updates bn.running_* in place"""
        k = _raw(3, c_st, dtype=torch.float32, device=dev)
        L.check(L.lib.mdie_bn_fold(c_st, c_real, split, gap, mean.data_ptr(), var.data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr(), EPS, momentum,
                                   count, bn.running_mean.data_ptr(), bn.running_var.data_ptr(), k[0].data_ptr(), k[1].data_ptr(), k[2].data_ptr(),
                                   _sp(dev)), "mdie_bn_fold")
        return k


def _bn_apply_inplace(dt, y, dz, k, mean, coef):
    """dz <- scale * (dz - k2 - xhat * k3)   (dz already masked: relu = 0)"""
    B, _, H, W = y.shape
    d = L.BnBwdDesc()
    d.dtype, d.N, d.nseg = dt, B * H * W, 1
    ptr, c, st = _nhwc(y)
    d.x[0] = L.Seg(ptr, c, st)
    ptr, c, st = _nhwc(dz)
    d.g[0] = L.Seg(ptr, c, st)
    d.accumulate = 0
    d.da, d.da_stride = ptr, st
    d.mean, d.invstd, d.scale, d.shift, d.relu = mean.data_ptr(), k[2].data_ptr(), k[0].data_ptr(), k[1].data_ptr(), 0
    d.coef = coef.data_ptr()
    L.check(L.lib.mdie_bn_bwd_apply(C.byref(d), _sp(y.device)), "mdie_bn_bwd_apply")


class _ConvBnFn(torch.autograd.Function):
    """(o, t) = (pool?(relu(bn(conv3x3(x) + b))), dropout(o)); o or t is None when not requested."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, bn, dt, pool, p, seed, need_o, need_t, seed_dev=None):
        B, cin_st, H, W = x.shape
        dev = x.device
        cout, cin = weight.shape[0], weight.shape[1]
        w32 = _f32(weight)
        y = _empty(dt, B, cout, H, W, dev)
        _conv_raw(dt, [x], _pack(dt, w32, 3, False, cout, cin, cout, cin_st), _f32(bias), 3, cout, y)
        mv = _raw(2, cout, dtype=torch.float32, device=dev)
        k = _Bn.stats_fold(dt, y, mv, 0, cout, cout, cout, 0, bn, 0.1)
        Ho, Wo = (H // 2, W // 2) if pool else (H, W)
        o = _empty(dt, B, cout, Ho, Wo, dev) if need_o else None
        t = _empty(dt, B, cout, Ho, Wo, dev) if need_t else None
        L.check(L.lib.mdie_bn_act_pool_fwd(dt, B, H, W, cout, y.data_ptr(), cout, k[0].data_ptr(), k[1].data_ptr(), int(pool),
                                           o.data_ptr() if need_o else None, cout, t.data_ptr() if need_t else None, cout, p, seed,
                                           seed_dev.data_ptr() if seed_dev is not None else None, _sp(dev)),
                "mdie_bn_act_pool_fwd")
        ctx.save_for_backward(x, w32, y, k, mv)
        ctx.set_materialize_grads(False)
        ctx.meta = (dt, pool, p, seed, cin, cout, need_o, need_t)
        ctx.wparam, ctx.bparam, ctx.gparams = weight, bias, (gamma, beta)
        ctx.seed_dev = seed_dev
        outs = tuple(v for v in (o, t) if v is not None)
        return outs if len(outs) > 1 else outs[0]

    @staticmethod
    def backward(ctx, *grads):
        dt, pool, p, seed, cin, cout, need_o, need_t = ctx.meta
        x, w32, y, k, mv = ctx.saved_tensors
        grads = list(grads)
        d_o = grads.pop(0) if need_o else None
        d_t = grads.pop(0) if need_t else None
        B, cin_st, H, W = x.shape
        dev = x.device
        td = E.TORCH_DTYPE[dt]
        d_o = _cl(d_o.to(td)) if d_o is not None else None
        d_t = _cl(d_t.to(td)) if d_t is not None else None
        dz = _raw_like(y)
        dgb = (_gout(ctx.gparams[0], (cout,), dev), _gout(ctx.gparams[1], (cout,), dev))
        coef = _raw(2, cout, dtype=torch.float32, device=dev)
        nws = L.lib.mdie_bn_workspace_bytes(cout)
        ws = _raw(nws, dtype=torch.uint8, device=dev)
        d = L.BnPoolBwdDesc()
        d.dtype, d.B, d.H, d.W, d.C, d.c_real = dt, B, H, W, cout, cout
        d.y, d.y_stride = y.data_ptr(), cout
        d.scale, d.shift, d.mean, d.invstd, d.pool = k[0].data_ptr(), k[1].data_ptr(), mv[0].data_ptr(), k[2].data_ptr(), int(pool)
        d.d_out, d.d_out_stride = (d_o.data_ptr(), cout) if d_o is not None else (None, 0)
        d.d_drop, d.d_drop_stride = (d_t.data_ptr(), cout) if d_t is not None else (None, 0)
        d.p, d.seed = p, seed
        d.seed_dev = ctx.seed_dev.data_ptr() if ctx.seed_dev is not None else None
        d.dz, d.dz_stride = dz.data_ptr(), cout
        d.dgamma, d.dbeta, d.coef = dgb[0].data_ptr(), dgb[1].data_ptr(), coef.data_ptr()
        d.workspace, d.workspace_bytes = ws.data_ptr(), nws
        d.two_pass = 1                                         # dz receives dL/dy itself (no mdie_bn_bwd_apply pass over it)
        L.check(L.lib.mdie_bn_act_pool_bwd(C.byref(d), _sp(dev)), "mdie_bn_act_pool_bwd")
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _empty(dt, B, cin_st, H, W, dev)
            _conv_raw(dt, [dz], _pack(dt, w32, 3, True, cin, cout, cin_st, cout), _zeros(cin_st, dev), 3, cin_st, dx)
        dw = _wgrad(dt, [x], dz, w32.shape, 3, False, cin, cout, cout, param=ctx.wparam)
        return dx, dw, _zero_grad_vec(cout, dev, ctx.bparam), dgb[0], dgb[1], None, None, None, None, None, None, None, None


class _DenseFn(torch.autograd.Function):
    """DenseBlock.forward (models/cdan.py:32-39) [+ torch.sigmoid, :157].  x: NHWC with `c0` stored channels of which
    `real_c` are real (3 of 16 for final_dense).  params: 4 x (bn.weight, bn.bias, conv.weight, conv.bias) + transition."""

    @staticmethod
    def forward(ctx, x, blk, dt, real_c, sigmoid, *params):
        B, c0, H, W = x.shape
        dev, N = x.device, B * H * W
        gap = c0 - real_c
        ct = c0 + 64
        # the four growth maps: one 16-channel tensor EACH (not slices of a 64-channel one): a pixel's 32 bytes of one map are then
        # contiguous with its neighbours', so the layer that writes it, the statistics pass and every later reader move whole
        # cache lines (a 16-of-64 slice touches a quarter of each 128-byte line: the statistics pass ran at 1.7 TB/s)
        grow = [_empty(dt, B, 16, H, W, dev) for _ in range(4)]
        mv = _raw(2, ct, dtype=torch.float32, device=dev)
        bn_of = lambda l: (getattr(blk.layers, str(l)) if l < 4 else blk.transition_layer)._modules["0"]
        # statistics once per segment (x, then each growth map as it is written), each pass followed in the same call by the
        # fold of the NEXT layer's BatchNorm over everything written so far
        k = _Bn.stats_fold(dt, x, mv, 0, c0, real_c, real_c, gap, bn_of(0), 0.1)
        consts, weights = [], []
        for l in range(5):
            w = _f32(params[4 * l + 2])      # (real input channels; `gap`: they sit at their stored positions, zero weights on the padding)
            cin_st, cin_real = c0 + 16 * l, real_c + 16 * l
            segs = [x] + grow[:l]
            if l < 4:
                out = grow[l]
                _conv_raw(dt, segs, _pack(dt, w, 3, False, 16, cin_real, 16, cin_st, split=real_c, gap=gap), _f32(params[4 * l + 3]), 3, 16, out, pre=(k[0], k[1]))
                consts.append(k)
                k = _Bn.stats_fold(dt, out, mv, cin_st, cin_st + 16, cin_real + 16, real_c, gap, bn_of(l + 1), 0.1)
            else:
                cout = w.shape[0]
                cout_st = (cout + 15) // 16 * 16
                packed = _pack(dt, w, 1, False, cout, cin_real, cout_st, cin_st, split=real_c, gap=gap)
                out = _empty(dt, B, cout_st, H, W, dev)
                y = _raw(B, 3, H, W, dtype=torch.float32, device=dev) if sigmoid else None
                _conv_raw(dt, segs, packed, _pad_vec(params[4 * l + 3], cout_st), 1, cout_st, out, pre=(k[0], k[1]),
                          act=L.ACT_SIGMOID if sigmoid else L.ACT_NONE, out_nchw3=y)
                consts.append(k)
            weights.append(w)
        ctx.save_for_backward(x, mv, *grow, *consts, *weights, *([y] if sigmoid else []))
        ctx.meta = (dt, real_c, sigmoid, c0)
        ctx.wparams = [params[4 * l + 2] for l in range(5)]
        ctx.params = params
        return y if sigmoid else out

    @staticmethod
    def backward(ctx, d_out):
        dt, real_c, sigmoid, c0 = ctx.meta
        saved = ctx.saved_tensors
        x, mv, grow = saved[0], saved[1], list(saved[2:6])
        consts, weights = saved[6:11], saved[11:16]
        B, _, H, W = x.shape
        dev, N, td = x.device, B * H * W, E.TORCH_DTYPE[dt]
        gap = c0 - real_c
        if sigmoid:
            y = saved[16]
            dz = _empty(dt, B, 16, H, W, dev)
            L.check(L.lib.mdie_sigmoid_bwd_nchw3(dt, B, H, W, _f32(d_out).data_ptr(), y.data_ptr(), dz.data_ptr(), 16, _sp(dev)), "mdie_sigmoid_bwd_nchw3")
        else:
            dz = _cl(d_out.to(td))
        gx, gg = _raw_like(x), [_raw_like(g) for g in grow]
        # 16-bit storage: a segment's gradient is the sum over its (up to five) consuming layers -- kept in fp32 until the
        # last consumer has added its part, then rounded ONCE into gx / gg (csrc/bn.hip bn_bwd_apply_kernel, acc32)
        acc32 = dt != L.F32 and ACC32
        if acc32:
            sx = _raw(x.shape, dtype=torch.float32, device=dev, memory_format=torch.channels_last)
            sg = [_raw(g.shape, dtype=torch.float32, device=dev, memory_format=torch.channels_last) for g in grow]
        grads = [None] * 20
        # The gradient of a feature segment (the block input x, a growth map g_s) is the sum of the BatchNorm-ReLU backward
        # terms of every layer that consumed it.  Each layer's da (gradient w.r.t. its activated input) is written ONE PLANE PER
        # 16 CHANNELS (csrc/conv_planar.hip), so a segment's slice of it is a dense stream, and the segment's gradient is formed
        # in ONE pass right before its producer needs it (mdie_bn_bwd_apply_multi: x, each consumer's plane and the sum move
        # once, one rounding).  Rounds 1-2 added the terms layer by layer into running sums over ALL segments (read x, da, sum;
        # write sum: 4 passes over c0 + 16 l channels per layer, 1.6 ms of a 10.9 ms step at 512x512, B = 8).
        esz = x.element_size()
        das, coefs = [None] * 5, [None] * 5
        for l in (4, 3, 2, 1, 0):
            w, k = weights[l], consts[l]
            cin_st = c0 + 16 * l
            cin_real = real_c + 16 * l
            segs = [x] + grow[:l]
            gsegs = [gx] + gg[:l]
            if l == 4:
                cout, cout_st, ks, dy = w.shape[0], dz.shape[1], 1, dz
                mean_dy = _raw(2, cout_st, dtype=torch.float32, device=dev)
                _Bn.stats(dt, dy, mean_dy[0], mean_dy[1])
                grads[4 * l + 3] = torch.mul(mean_dy[0, :cout], N, out=_gout(ctx.params[4 * l + 3], (cout,), dev))   # the only bias here that is not followed by a BatchNorm
            else:
                cout, cout_st, ks, dy = 16, 16, 3, gg[l]   # complete: every consumer of this segment has run
                grads[4 * l + 3] = _zero_grad_vec(16, dev, ctx.params[4 * l + 3])
            # gradient w.r.t. the activated input a = relu(bn(cat(segs)))
            planar = not acc32
            fuse = planar and BN_REDUCE_IN_DGRAD
            da = _raw(cin_st // 16, N, 16, dtype=td, device=dev) if planar else _empty(dt, B, cin_st, H, W, dev)
            dgb = (_gout(ctx.params[4 * l], (cin_real,), dev), _gout(ctx.params[4 * l + 1], (cin_real,), dev))
            coef = _raw(2, cin_st, dtype=torch.float32, device=dev)
            if fuse:    # the BatchNorm-ReLU backward sums come out of the input-gradient convolution's epilogue (mdie_conv_desc.bnred)
                nslab = L.lib.mdie_conv_bnred_slabs(B, H, W, cin_st)
                partial = _raw(nslab, 2, cin_st, dtype=torch.float32, device=dev)
                bnred = (segs, k[0], k[1], partial)
            else:
                bnred = None
            _conv_raw(dt, [dy], _pack(dt, w, ks, True, cin_real, cout, cin_st, cout_st, out_split=real_c, out_gap=gap), _zeros(cin_st, dev), ks, cin_st, da, planar=planar,
                      bnred=bnred)
            grads[4 * l + 2] = _wgrad(dt, segs, dy, (cout, cin_real, ks, ks), ks, False, cin_real, cout, cout_st, pre=(k[0], k[1]), split=real_c, gap=gap,
                                      param=ctx.wparams[l])
            # BatchNorm + ReLU backward: per-channel sums of this layer ...
            d = L.BnBwdDesc()
            if fuse:
                f = L.BnBwdFinishDesc()
                f.C, f.N, f.partial, f.n_partial = cin_st, N, partial.data_ptr(), nslab
                f.mean, f.invstd = mv[0].data_ptr(), k[2].data_ptr()
                f.c_real, f.split, f.gap = cin_real, real_c, gap
                f.dgamma, f.dbeta, f.coef = dgb[0].data_ptr(), dgb[1].data_ptr(), coef.data_ptr()
                L.check(L.lib.mdie_bn_bwd_finish(C.byref(f), _sp(dev)), "mdie_bn_bwd_finish")
            else:
                nws = L.lib.mdie_bn_workspace_bytes(cin_st)
                ws = _raw(nws, dtype=torch.uint8, device=dev)
                d.dtype, d.N, d.nseg = dt, N, len(segs)
                for i, (sgm, g) in enumerate(zip(segs, gsegs)):
                    ptr, c, st = _nhwc(sgm)
                    d.x[i] = L.Seg(ptr, c, st)
                    ptr, c, st = _nhwc(g)
                    d.g[i] = L.Seg(ptr, c, st)
                d.accumulate = 0 if l == 4 else 31
                if acc32:
                    ptr, c, st = _nhwc(sx)
                    d.acc32[0], d.final_from[0] = L.Seg(ptr, c, st), (0 if l == 0 else c0)        # x: layer 0 is its last consumer
                    for j in range(l):
                        ptr, c, st = _nhwc(sg[j])
                        d.acc32[1 + j], d.final_from[1 + j] = L.Seg(ptr, c, st), (0 if j == l - 1 else 16)   # growth map l-1: this layer is its last consumer
                if planar:
                    d.da, d.da_stride, d.da_plane = da.data_ptr(), 16, N * 16
                else:
                    d.da, d.da_stride = da.data_ptr(), cin_st
                d.mean, d.invstd, d.scale, d.shift, d.relu = mv[0].data_ptr(), k[2].data_ptr(), k[0].data_ptr(), k[1].data_ptr(), 1
                d.c_real, d.split, d.gap = cin_real, real_c, gap
                d.dgamma, d.dbeta, d.coef = dgb[0].data_ptr(), dgb[1].data_ptr(), coef.data_ptr()
                d.workspace, d.workspace_bytes = ws.data_ptr(), nws
                L.check(L.lib.mdie_bn_bwd_reduce(C.byref(d), _sp(dev)), "mdie_bn_bwd_reduce")
            grads[4 * l], grads[4 * l + 1] = dgb[0], dgb[1]
            if acc32:   # (fp32 running sums: the layer-by-layer form)
                L.check(L.lib.mdie_bn_bwd_apply(C.byref(d), _sp(dev)), "mdie_bn_bwd_apply")
                continue
            das[l], coefs[l] = da, coef
            # ... then the gradient of the segment whose last consumer has now run: growth map l - 1 (consumers l .. 4), or,
            # after layer 0, the block input (all five)
            if l == 0 and not ctx.needs_input_grad[0]:
                continue
            off, Cs = (c0 + 16 * (l - 1), 16) if l else (0, c0)
            xs, gs = (grow[l - 1], gg[l - 1]) if l else (x, gx)
            m = L.BnBwdMultiDesc()
            m.dtype, m.N, m.C = dt, N, Cs
            ptr, c, st = _nhwc(xs)
            m.x, m.x_stride = ptr, st
            ptr, c, st = _nhwc(gs)
            m.g, m.g_stride = ptr, st
            m.mean, m.invstd = mv[0, off:].data_ptr(), consts[4][2, off:].data_ptr()    # (the same batch statistics in every layer's constants)
            layers = range(l, 5)
            m.nlayer = len(layers)
            for i, j in enumerate(layers):
                m.da[i], m.da_stride[i], m.da_plane[i] = das[j].data_ptr() + (off // 16) * N * 16 * esz, 16, N * 16
                m.scale[i], m.shift[i] = consts[j][0, off:].data_ptr(), consts[j][1, off:].data_ptr()
                m.coef[i], m.coef_stride[i] = coefs[j][0, off:].data_ptr(), c0 + 16 * j
            L.check(L.lib.mdie_bn_bwd_apply_multi(C.byref(m), _sp(dev)), "mdie_bn_bwd_apply_multi")
        return (gx if ctx.needs_input_grad[0] else None, None, None, None, None, *grads)


class _DeconvFn(torch.autograd.Function):
    """out = up2x?(relu(bn(conv_transpose3x3(x) + b))) + skip  (models/cdan.py:127-130,134-138,142-146,150-154)"""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, skip, bn, dt, up):
        B, cin, H, W = x.shape
        dev = x.device
        cout = weight.shape[1]
        cout_st = (cout + 15) // 16 * 16
        w32 = _f32(weight)
        y = _empty(dt, B, cout_st, H, W, dev)
        _conv_raw(dt, [x], _pack(dt, w32, 3, True, cout, cin, cout_st, cin), _pad_vec(bias, cout_st), 3, cout_st, y)
        mv = _raw(2, cout_st, dtype=torch.float32, device=dev)
        k = _Bn.stats_fold(dt, y, mv, 0, cout_st, cout, cout_st, 0, bn, 0.1)
        Ho, Wo = (2 * H, 2 * W) if up else (H, W)
        out = _empty(dt, B, cout_st, Ho, Wo, dev)
        sptr, sc, sst = _nhwc(skip)
        assert sc == cout_st and skip.shape[2] == Ho
        L.check(L.lib.mdie_bn_act_up_add_fwd(dt, B, H, W, cout_st, y.data_ptr(), cout_st, k[0].data_ptr(), k[1].data_ptr(), int(up), sptr, sst,
                                             out.data_ptr(), cout_st, _sp(dev)), "mdie_bn_act_up_add_fwd")
        ctx.save_for_backward(x, w32, y, k, mv)
        ctx.meta = (dt, up, cin, cout, cout_st)
        ctx.wparam, ctx.bparam, ctx.gparams = weight, bias, (gamma, beta)
        return out

    @staticmethod
    def backward(ctx, d_out):
        dt, up, cin, cout, cout_st = ctx.meta
        x, w32, y, k, mv = ctx.saved_tensors
        B, _, H, W = x.shape
        dev, td = x.device, E.TORCH_DTYPE[dt]
        d_out = _cl(d_out.to(td))
        dz = _raw_like(y)
        dgb = (_gout(ctx.gparams[0], (cout,), dev), _gout(ctx.gparams[1], (cout,), dev))
        coef = _raw(2, cout_st, dtype=torch.float32, device=dev)
        nws = L.lib.mdie_bn_workspace_bytes(cout_st)
        ws = _raw(nws, dtype=torch.uint8, device=dev)
        if up:
            d = L.BnUpBwdDesc()
            d.dout, d.dout_stride = d_out.data_ptr(), cout_st
        else:
            d = L.BnPoolBwdDesc()
            d.pool, d.d_out, d.d_out_stride, d.d_drop, d.d_drop_stride, d.p, d.seed = 0, d_out.data_ptr(), cout_st, None, 0, 0.0, 0
        d.dtype, d.B, d.H, d.W, d.C, d.c_real = dt, B, H, W, cout_st, cout
        d.y, d.y_stride = y.data_ptr(), cout_st
        d.scale, d.shift, d.mean, d.invstd = k[0].data_ptr(), k[1].data_ptr(), mv[0].data_ptr(), k[2].data_ptr()
        d.dz, d.dz_stride = dz.data_ptr(), cout_st
        d.dgamma, d.dbeta, d.coef = dgb[0].data_ptr(), dgb[1].data_ptr(), coef.data_ptr()
        d.workspace, d.workspace_bytes = ws.data_ptr(), nws
        if up:
            L.check(L.lib.mdie_bn_act_up_bwd(C.byref(d), _sp(dev)), "mdie_bn_act_up_bwd")
        else:
            L.check(L.lib.mdie_bn_act_pool_bwd(C.byref(d), _sp(dev)), "mdie_bn_act_pool_bwd")
        _bn_apply_inplace(dt, y, dz, k, mv[0], coef)
        dx = _empty(dt, B, cin, H, W, dev)
        # input gradient of a transposed convolution = plain convolution with the un-flipped kernel, in/out swapped
        _conv_raw(dt, [dz], _pack(dt, w32, 3, False, cin, cout, cin, cout_st), _zeros(cin, dev), 3, cin, dx)
        dw = _wgrad(dt, [x], dz, w32.shape, 3, True, cin, cout, cout_st, param=ctx.wparam)
        return dx, dw, _zero_grad_vec(cout, dev, ctx.bparam), dgb[0], dgb[1], (d_out if ctx.needs_input_grad[5] else None), None, None, None


# ---- the generic convolution Function (kept for callers that compose their own blocks and for the operator tests) -------
class _ConvFn(torch.autograd.Function):
    """y = conv_k(cat(segments), weight) + bias on the HIP engine; stored channel counts are multiples of 16.
    weight: [cout, cin, k, k] (transposed=False) or [cin, cout, k, k] (transposed=True, ConvTranspose2d k3 s1 p1)."""

    @staticmethod
    def forward(ctx, dtype, transposed, weight, bias, *segs):
        ks = weight.shape[2]
        cout, cin = (weight.shape[1], weight.shape[0]) if transposed else (weight.shape[0], weight.shape[1])
        assert cin == sum(s.shape[1] for s in segs) and cin % 16 == 0 and cout % 16 == 0
        w32 = _f32(weight)
        B, _, H, W = segs[0].shape
        out = _empty(dtype, B, cout, H, W, weight.device)
        _conv_raw(dtype, list(segs), _pack(dtype, w32, ks, transposed, cout, cin), _f32(bias), ks, cout, out)
        ctx.save_for_backward(w32, *segs)
        ctx.meta = (dtype, transposed, ks, cout, cin)
        ctx.wparam = weight
        return out

    @staticmethod
    def backward(ctx, dy):
        dtype, transposed, ks, cout, cin = ctx.meta
        w32, *segs = ctx.saved_tensors
        dy = _cl(dy.to(E.TORCH_DTYPE[dtype]))
        B, _, H, W = dy.shape
        dev = dy.device
        # input gradient: the same convolution with the in/out-swapped, flipped kernel
        dx = _empty(dtype, B, cin, H, W, dev)
        _conv_raw(dtype, [dy], _pack(dtype, w32, ks, not transposed, cin, cout), _zeros(cin, dev), ks, cin, dx)
        dw = _wgrad(dtype, list(segs), dy, w32.shape, ks, transposed, cin, cout, cout, param=ctx.wparam)
        db = dy.float().sum(dim=(0, 2, 3))
        grads, c0 = [], 0
        for s in segs:
            grads.append(dx[:, c0:c0 + s.shape[1]])
            c0 += s.shape[1]
        return (None, None, dw, db, *grads)


def conv(dtype, weight, bias, segs, transposed=False):
    return _ConvFn.apply(dtype, transposed, weight, bias, *[_cl(s) for s in segs])


def _step_tensors(net):
    """(the num_batches_tracked buffers, the 4-D fp32 parameters) of `net`, found once: walking the module tree costs 0.7 ms of an
    8 ms step.  Valid while the first and last of each list are still the tensors the modules hold (load_state_dict copies in
    place and keeps them; .to() / a re-wrap replaces them -- then the lists are rebuilt)."""
    c = net.__dict__.get("_mdie_step_tensors")
    if c is not None:
        bufs, convs, probes = c
        if all(getattr(m, n, None) is t for m, n, t in probes):
            return bufs, convs
    bufs, convs, probes = [], [], []
    for m in net.modules():
        for n, b in m._buffers.items():
            if n == "num_batches_tracked" and b is not None:
                bufs.append(b)
                probes.append((m, n, b))
        for n, p in m._parameters.items():
            if p is not None and p.dim() == 4 and p.dtype == torch.float32 and p.is_contiguous():
                convs.append(p)
                probes.append((m, n, p))
    net.__dict__["_mdie_step_tensors"] = (bufs, convs, probes)
    return bufs, convs


def _tick(net):
    bufs = _step_tensors(net)[0]
    if bufs:
        torch._foreach_add_(bufs, 1)        # (one or two launches instead of 32)


def _dropout_state(net, dev):
    """(salt, step counter on the device).  The masks of a step are hash(salt + layer, counter, element): the counter
    lives in device memory and is bumped by a (capturable) kernel once per forward, so a step replayed from a hipGraph
    draws new masks although every host-side argument of its launches is frozen; the salt comes from torch's CPU
    generator once per network (torch.manual_seed reproduces a run)."""
    st = getattr(net, "_mdie_dropout", None)
    if st is None or st[1].device != dev:
        st = (int(torch.randint(0, 2 ** 31 - 1, (1,)).item()), torch.zeros(1, dtype=torch.int32, device=dev))
        net._mdie_dropout = st
    return st


def _dense_params(blk):
    ps = []
    for i in range(4):
        seq = getattr(blk.layers, str(i))._modules
        ps += [seq["0"].weight, seq["0"].bias, seq["2"].weight, seq["2"].bias]
    seq = blk.transition_layer._modules
    return ps + [seq["0"].weight, seq["0"].bias, seq["2"].weight, seq["2"].bias]


def dense_block(dt, blk, x, real_c, sigmoid=False):
    return _DenseFn.apply(x, blk, dt, real_c, sigmoid, *_dense_params(blk))


def conv_block(dt, blk, x, pool, p, need_o=True, need_t=True, seed=None, seed_dev=None):
    if seed is None:
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if p > 0 else 0       # CPU generator: no device sync
    return _ConvBnFn.apply(x, blk.conv.weight, blk.conv.bias, blk.bn.weight, blk.bn.bias, blk.bn, dt, pool, float(p), seed, need_o, need_t, seed_dev)


def deconv_stage(dt, cv, bn, x, skip, up):
    return _DeconvFn.apply(x, cv.weight, cv.bias, bn.weight, bn.bias, skip, bn, dt, up)




class _CbamFn(torch.autograd.Function):
    """CBAM.forward (models/cbam.py:91-95) in training mode, optionally times `mul` (the `out *= dense_k` that follows it)."""

    @staticmethod
    def forward(ctx, x, mul, w1, b1, w2, b2, w7, gamma, beta, bn, dt):
        B, Cc, H, W = x.shape
        dev = x.device
        f32 = lambda *shape: _raw(*shape, dtype=torch.float32, device=dev)
        gate, pooled, comp, smap, bnc = f32(B, Cc), f32(B, 2, Cc), f32(B, H, W, 2), f32(B, H, W), f32(4)
        amax = _raw(B, Cc, dtype=torch.int32, device=dev)
        out = _raw_like(x)
        params = [_f32(p) for p in (w1, b1, w2, b2, w7, gamma, beta)]
        d = _CbamFn._desc(dt, x, mul, params, gate, amax, pooled, comp, smap, bnc)
        d.out, d.out_stride = out.data_ptr(), out.stride(3)
        d.running_mean, d.running_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
        nws = L.lib.mdie_cbam_train_workspace_bytes(B, H, W, Cc)
        ws = _raw(nws, dtype=torch.uint8, device=dev)
        d.workspace, d.workspace_bytes = ws.data_ptr(), nws
        L.check(L.lib.mdie_cbam_train_fwd(C.byref(d), _sp(dev)), "mdie_cbam_train_fwd")
        ctx.save_for_backward(x, mul, gate, amax, pooled, comp, smap, bnc, *params)
        ctx.dt = dt
        ctx.pparams = (w1, b1, w2, b2, w7, gamma, beta)
        return out

    @staticmethod
    def _desc(dt, x, mul, params, gate, amax, pooled, comp, smap, bnc):
        B, Cc, H, W = x.shape
        d = L.CbamTrainDesc()
        d.dtype, d.B, d.H, d.W, d.C = dt, B, H, W, Cc
        ptr, _, st = _nhwc(x)
        d.x, d.x_stride = ptr, st
        if mul is not None:
            ptr, _, st = _nhwc(mul)
            d.mul, d.mul_stride = ptr, st
        d.w1, d.b1, d.w2, d.b2, d.w7, d.gamma, d.beta = [p.data_ptr() for p in params]
        d.momentum, d.eps = 0.01, EPS
        d.gate, d.amax_idx, d.pooled, d.comp, d.smap, d.bnc = [t.data_ptr() for t in (gate, amax, pooled, comp, smap, bnc)]
        return d

    @staticmethod
    def backward(ctx, d_out):
        x, mul, gate, amax, pooled, comp, smap, bnc, *params = ctx.saved_tensors
        dt = ctx.dt
        B, Cc, H, W = x.shape
        dev = x.device
        d_out = _cl(d_out.to(E.TORCH_DTYPE[dt]))
        d = _CbamFn._desc(dt, x, mul, params, gate, amax, pooled, comp, smap, bnc)
        dx = _raw_like(x)
        dmul = _raw_like(mul) if mul is not None else None
        grads = [_gout(pp, p.shape, dev) for pp, p in zip(ctx.pparams, params)]
        d.dout, d.dout_stride = d_out.data_ptr(), d_out.stride(3)
        d.dx, d.dx_stride = dx.data_ptr(), dx.stride(3)
        if dmul is not None:
            d.dmul, d.dmul_stride = dmul.data_ptr(), dmul.stride(3)
        d.dw1, d.db1, d.dw2, d.db2, d.dw7, d.dgamma, d.dbeta = [g.data_ptr() for g in grads]
        nws = L.lib.mdie_cbam_train_workspace_bytes(B, H, W, Cc)
        ws = _raw(nws, dtype=torch.uint8, device=dev)
        d.workspace, d.workspace_bytes = ws.data_ptr(), nws
        L.check(L.lib.mdie_cbam_train_bwd(C.byref(d), _sp(dev)), "mdie_cbam_train_bwd")
        return (dx, dmul, *grads, None, None)


def cbam(dt, node, x, mul=None):
    mlp, sp = node.ChannelGate.mlp._modules, node.SpatialGate.spatial
    return _CbamFn.apply(_cl(x), None if mul is None else _cl(mul), mlp["1"].weight, mlp["1"].bias, mlp["3"].weight, mlp["3"].bias,
                         sp.conv.weight, sp.bn.weight, sp.bn.bias, sp.bn, dt)


def forward_train(net, x, precision="fp32", dropout_p=0.2):
    """CDAN.forward (models/cdan.py:171-176) with the module in training mode.  x: fp32 NCHW on the GPU."""
    if not x.is_cuda:
        raise L.MdieError("forward_train: GPU tensors only (no CPU fallback)")
    dt = E.dtype_id(precision)
    enc, dec = net.encoder, net.decoder
    _tick(net)
    global _PLAN
    plans = net.__dict__.setdefault("_mdie_pack_plans", {})
    _PLAN = plans.get(dt) or plans.setdefault(dt, _PackPlan(dt))
    ptrs = frozenset(p.data_ptr() for p in _step_tensors(net)[1])
    if ptrs != _PLAN.param_ptrs:        # parameters moved (load_state_dict keeps them; .to() / a new optimizer wrapper may not): start over
        plans[dt] = _PLAN = _PackPlan(dt)
        _PLAN.param_ptrs = ptrs
    _PLAN.run(x.device)
    _new_zero_arena(x.device)
    B, ch, H, W = x.shape
    global _wgrad_side_this_step
    _wgrad_side_this_step = WGRAD_STREAM and B * H * W >= WGRAD_STREAM_MIN_PIXELS and not torch.cuda.is_current_stream_capturing()
    if ch != 3 or H % 8 or W % 8:
        raise L.MdieError(f"forward_train: input must be [B,3,H,W] with H, W multiples of 8, got {tuple(x.shape)}")
    xin = _empty(dt, B, 16, H, W, x.device)
    L.check(L.lib.mdie_nchw3_to_nhwc16(dt, B, H, W, _f32(x).data_ptr(), xin.data_ptr(), _sp(x.device)), "mdie_nchw3_to_nhwc16")
    salt, counter = (0, None)
    if dropout_p > 0:
        salt, counter = _dropout_state(net, x.device)
        counter += 1                                  # on the device: a new set of masks per step, also under graph replay
    t = xin
    skips, denses = [], []
    for i in (1, 2, 3):
        if dropout_p > 0:
            o, t = conv_block(dt, getattr(enc, f"conv{i}"), t, True, dropout_p, seed=(salt + 7919 * i) & 0x7fffffff, seed_dev=counter)
        else:
            o = t = conv_block(dt, getattr(enc, f"conv{i}"), t, True, 0.0, need_t=False)
        denses.append(dense_block(dt, getattr(enc, f"dense{i}"), o, o.shape[1]))
        skips.append(t)
    e = conv_block(dt, enc.conv4, t, False, dropout_p, need_o=False, seed=(salt + 7919 * 4) & 0x7fffffff, seed_dev=counter)
    t = cbam(dt, net.bottleneck, e)
    t = deconv_stage(dt, dec.conv1, dec.bn1, t, skips[2], False)
    t = cbam(dt, dec.cbam1, t, denses[2])
    t = deconv_stage(dt, dec.conv2, dec.bn2, t, skips[1], True)
    t = cbam(dt, dec.cbam2, t, denses[1])
    t = deconv_stage(dt, dec.conv3, dec.bn3, t, skips[0], True)
    t = cbam(dt, dec.cbam3, t, denses[0])
    t = deconv_stage(dt, dec.conv4, dec.bn4, t, xin, True)
    return dense_block(dt, dec.final_dense, t, 3, sigmoid=True)


# ---- one training step as a hipGraph (models/model.py:154-172: zero_grad, forward, loss, backward, optimizer step) -------------------
ALLOW_IN_GRAPH_EXCHANGE = False     # tests / experiments: CapturedStep(buckets=...) without the environment switch
_CAPTURING_STEP = None     # the CapturedStep whose graph is being recorded right now (GradBuckets._launch checks the stream against it)


class CapturedStep:
    """forward + loss + backward (+ the optimizer step) of ONE batch shape, captured once and replayed.

    An eager step is ~520 kernel launches issued from Python through autograd: below 512x512 the GPU waits for the host.
    Everything on the step is capturable -- the HIP entry points neither allocate nor synchronise, tensors come from
    torch's graph-private pool, the dropout counter and BatchNorm's num_batches_tracked are bumped on the device -- so the
    whole step becomes one graph launch.  `optimizer` must have been built with capturable=True (Adam keeps its step
    count on the device then); pass optimizer=None to capture forward + loss + backward only (gradient all-reduce and a
    GradScaler-guarded step then follow the replay in the usual way).

        step = CapturedStep(net, losses, opt, x0, t0)          # eager warm-up on a side stream, then capture
        values = step(x, t)                                     # device tensor [terms..., total]; ONE graph launch

    Building it leaves the training state untouched: the warm-up passes run forward + backward only and the BatchNorm /
    dropout buffers they advanced are put back; an optimizer WITHOUT state gets it from one step on all-zero gradients (no
    parameter moves) whose step count is reset -- an optimizer that has already stepped (a second batch shape captured in
    the middle of training: the last, partial batch of an epoch) is left exactly as it is.  `scale_fn` (e.g.
    GradScaler.scale) is applied to the total loss before backward inside the graph -- the scaler's factor is a device
    tensor, so replays follow its updates.

    Data parallel (`buckets`: a GradBuckets) -- UNSAFE, opt-in (MDIE_DDP_CAPTURE=1 | auto, or train.ALLOW_IN_GRAPH_EXCHANGE): a process that
    captures RCCL collectives this way can be ABORTED by torch's own NCCL watchdog thread.  Round 6 caught the text of what round 5 saw twice
    without one: `Process group watchdog thread terminated with exception: HIP error: operation not permitted on an event last recorded in
    a capturing stream` (hipErrorCapturedEvent), raised from `WorkNCCL::isCompleted()` <- `Watchdog::runLoop()`, then `terminate` -> SIGABRT
    (profiles/r06zz_nccl_watchdog_abort_in_graph_exchange.log: tools/bench_train.py's in-graph variant, 0.7 s after the group was created).
    The watchdog polls the end events of the Work objects it has been handed; one of them had been recorded while the communication stream
    was part of a capture.  Which Work that is lies inside torch (a plain probe -- eager all-reduce, then a capture with all-reduces, watchdog
    polling throughout -- does not reproduce it: tools/probe_nccl_capture.py, profiles/r06k_probe_nccl_capture.txt; here the collectives are
    issued from autograd's device thread by the buckets' hooks), it is a race against the watchdog's 100 ms poll, and nothing on this side
    of torch's API can order it.  The default data-parallel step is therefore EAGER with the collectives issued by the hooks; this form stays
    for whoever measures it on a torch that keeps captured Work objects away from the watchdog.
    The gradient exchange is INSIDE the graph.  The buckets' post-accumulate hooks fire while
    the backward is captured, so each bucket's all-reduce is recorded at the point of the backward where its last gradient has been
    produced -- `torch.distributed`'s NCCL (= RCCL) process group forks its communication stream from the capturing stream there -- and
    `finish()` records the join in front of the optimizer step: every replay overlaps the collectives with the rest of backward exactly as
    an eager hook-driven step does, with no host work at all.  The eager warm-up steps run the same collectives (every rank builds its
    CapturedStep at the same batch, so the calls match up across ranks).

    Gradients: the captured backward writes into gradient tensors of the graph's private pool (or into the bucket slices).  Each CapturedStep keeps
    its own (`self.grads`) and re-points every `p.grad` at them after a replay, so whatever reads `p.grad` next -- the
    gradient exchange, `GradScaler.step`, an eager optimizer step -- sees THIS replay's gradients no matter which other
    shape was captured or which eager step (`zero_grad(set_to_none=True)`) ran in between."""

    def __init__(self, net, losses, optimizer, x, t, warmup=2, scale_fn=None, buckets=None):
        dev = x.device
        self.net, self.losses, self.opt = net, losses, optimizer
        self.x, self.t = x.clone(), t.clone()
        self.params = [p for p in net.parameters() if p.requires_grad]
        self.buckets = buckets
        if buckets is not None:
            if not (ALLOW_IN_GRAPH_EXCHANGE or __import__("os").environ.get("MDIE_DDP_CAPTURE", "0") in ("1", "auto")):
                raise RuntimeError("CapturedStep(buckets=...): capturing the gradient exchange is opt-in (MDIE_DDP_CAPTURE=1 | auto): torch's NCCL watchdog "
                                   "can abort a process that does it (see the class docstring); the default is eager steps with the exchange from the hooks")
            buckets.attach()       # the hooks fire while the backward is being CAPTURED: each bucket's collective becomes a branch of the graph

        def fwd_bwd():
            total, values = losses(net(self.x), self.t)
            (scale_fn(total) if scale_fn is not None else total).backward()
            join_weight_gradients(dev)
            if buckets is not None:
                buckets.finish()   # the capturing stream joins the communication stream; .grad = the averaged slices
            return values

        targets = list(net.buffers())           # what a training-mode forward advances besides the parameters
        if getattr(net, "dropout_p", 0.0) > 0:
            targets.append(_dropout_state(net, dev)[1])
        saved = [b.detach().clone() for b in targets]
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                net.zero_grad(set_to_none=True)
                fwd_bwd()
            if optimizer is not None and len(optimizer.state) == 0:
                # create the optimizer state outside the capture without moving a parameter (fresh optimizers only: with
                # populated moments a zero-gradient step WOULD move every parameter and decay both moments)
                for p in self.params:
                    if p.grad is not None:
                        p.grad.zero_()
                optimizer.step()
                for st in optimizer.state.values():
                    if torch.is_tensor(st.get("step")):
                        st["step"].zero_()
            for dst, src in zip(targets, saved):
                dst.copy_(src)
        torch.cuda.current_stream(dev).wait_stream(side)
        net.zero_grad(set_to_none=True)
        self.graph = torch.cuda.CUDAGraph()
        global _CAPTURING_STEP
        _CAPTURING_STEP = self
        try:
            with torch.cuda.graph(self.graph):
                self.values = fwd_bwd()
                if optimizer is not None:
                    optimizer.step()
        finally:
            _CAPTURING_STEP = None
        self.grads = [p.grad for p in self.params]     # the graph's own gradient tensors (rewritten by every replay)

    def __call__(self, x, t):
        self.x.copy_(x, non_blocking=True)
        self.t.copy_(t, non_blocking=True)
        self.graph.replay()
        for p, g in zip(self.params, self.grads):
            p.grad = g
        if self.opt is not None and hasattr(self.net, "_mdie_epoch"):
            self.net._mdie_epoch += 1                  # parameters rewritten without a version bump: modules._fingerprint
        return self.values


# ---- data-parallel gradient exchange ---------------------------------------------------------------------
class GradBuckets:
    """Bucketed gradient all-reduce overlapped with backward; gradients live IN the buckets.

    Parameters are grouped in REVERSE registration order (gradients become ready roughly back-to-front) into `n_buckets` flat
    fp32 buckets of about equal size (the whole model is 14.3 MB: PyTorch-DDP's 25 MB default would be ONE bucket and zero
    overlap).  While an instance is active (`_GRAD_SINK`), the training Functions write every parameter gradient straight into
    its slice of its bucket and hand autograd a VIEW of it (`_gout`): `.grad` is bucket memory, nothing is copied in or out
    (round 3 paid 140 `copy_` into the buckets from the hooks and 140 back in `finish()`: 280 launches on a host-bound step).
    A post-accumulate-grad hook only COUNTS; when the last gradient of a bucket has arrived the bucket's all-reduce is launched
    asynchronously (RCCL over xGMI under the "nccl" backend, averaging in the collective itself: ReduceOp.AVG), so it runs
    under the rest of backward.  `finish()` waits (the current stream waits for the communication stream) and points every
    `.grad` at its slice.  A gradient that did not come from a sink-aware producer (any other autograd node, an existing
    `.grad` that autograd accumulated into) is copied into its slice by the hook -- the old path, still correct.

    The weight-gradient side stream (train._wgrad, opt-in, EXPERIMENTAL: its round-3 finding has no cause) is never taken while a
    GradBuckets is active: a data-parallel step runs the single-stream schedule.

    `exchange()` serves steps whose backward fired no hooks (a replayed CapturedStep): the captured kernels have written into
    the same slices (the sink was active at capture time), so it is the five all-reduces and nothing else."""

    def __init__(self, params, process_group=None, n_buckets=4):
        import torch.distributed as dist
        self.dist, self.group = dist, process_group
        self.world = dist.get_world_size(process_group)
        params = [p for p in params if p.requires_grad][::-1]
        total = sum(p.numel() for p in params)
        target = max(1, -(-total // n_buckets))
        self.buckets, cur, size = [], [], 0
        for p in params:
            if cur and size + p.numel() > 1.25 * target:   # a large tensor (decoder.conv1, encoder.conv4: 1.18 M each) starts its own
                self.buckets.append(cur)                   # bucket: the small ones before it are exchanged as soon as they are ready
                cur, size = [], 0
            cur.append(p)
            size += p.numel()
            if size >= target:
                self.buckets.append(cur)
                cur, size = [], 0
        if cur:
            self.buckets.append(cur)
        self.flat = [torch.zeros(sum(p.numel() for p in b), dtype=torch.float32, device=b[0].device) for b in self.buckets]
        self._where = {}
        for bi, b in enumerate(self.buckets):
            off = 0
            for p in b:
                self._where[p] = (bi, off)
                off += p.numel()
        backend = dist.get_backend(process_group)
        self._avg_op = dist.ReduceOp.AVG if backend == "nccl" else None       # (gloo has no AVG: sum, then one divide per bucket)
        self._dirty = set()           # parameters whose slice a kernel has written: no longer a valid source of exact zeros
        self._pending = [len(b) for b in self.buckets]
        self._work = [None] * len(self.buckets)
        self._hooks = []
        self._claim_task, self._claimed = None, set()
        self.trace = None             # tests: a list that receives ("grad", bucket) per arriving gradient and ("launch", bucket) per collective issued
        self.copies_in = 0            # gradients that had to be copied into their slice (diagnostic: 0 on the engine's own training path)
        self.activate()
        self.attach()

    # ---- the sink ----
    def activate(self):
        global _GRAD_SINK
        _GRAD_SINK = self

    def close(self):
        global _GRAD_SINK
        self.remove()
        if _GRAD_SINK is self:
            _GRAD_SINK = None

    def claim(self, p):
        """True the first time parameter `p` asks for its slice inside the running backward (one autograd graph task)"""
        task = _graph_task()
        if task is None or task < 0:
            return False
        if self._claim_task != task:
            self._claim_task, self._claimed = task, set()
        if id(p) in self._claimed:
            return False
        self._claimed.add(id(p))
        return True

    def _slice(self, p):
        w = self._where.get(p)
        if w is None:
            return None
        return self.flat[w[0]][w[1]:w[1] + p.numel()]

    def view_of(self, p):
        v = self._slice(p)
        if v is None:
            return None
        self._dirty.add(p)
        return v.view(p.shape)

    def zero_view_of(self, p):
        v = self._slice(p)
        if v is None:
            return None
        if p in self._dirty:          # a kernel wrote this slice in some earlier step (the parameter changed roles): make it zero again
            v.zero_()
            self._dirty.discard(p)
        return v.view(p.shape)

    def is_view(self, p, g):
        w = self._where.get(p)
        return w is not None and g is not None and g.data_ptr() == self.flat[w[0]].data_ptr() + 4 * w[1] and g.is_contiguous() and g.dtype == torch.float32

    # ---- overlap mode: hooks ----
    def attach(self):
        if not self._hooks:
            self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for b in self.buckets for p in b]
            global _wgrad_hooks_active
            _wgrad_hooks_active += 1      # (train._wgrad: the side stream is then taken only for parameters whose gradient is a bucket view)

    def remove(self):
        """stop reacting to backward (a replayed CapturedStep fires no hooks: `exchange()` follows the replay); the sink stays"""
        global _wgrad_hooks_active
        if self._hooks:
            _wgrad_hooks_active = max(0, _wgrad_hooks_active - 1)
        for h in self._hooks:
            h.remove()
        self._hooks = []

    def _launch(self, bi):
        if self.trace is not None:
            self.trace.append(("launch", bi))
        flat = self.flat[bi]
        if flat.is_cuda and torch.cuda.is_current_stream_capturing() != (_CAPTURING_STEP is not None):
            # a hook that fires on a thread / stream other than the one CapturedStep is capturing on would issue its collective OUTSIDE the
            # graph (or an eager step would record one into somebody's capture): refuse loudly rather than exchange stale gradients
            raise RuntimeError("GradBuckets: a bucket's all-reduce was about to be issued on a stream whose capture state does not match the step "
                               "being built (CapturedStep capturing: %s, current stream capturing: %s)" % (_CAPTURING_STEP is not None, torch.cuda.is_current_stream_capturing()))
        side = _WGRAD_SIDE.get(flat.device) if (flat.is_cuda and _wgrad_side_this_step and WGRAD_STREAM) else None
        if side is not None:          # dW kernels of this bucket may be on the side stream: order the collective behind both streams
            side.wait_stream(torch.cuda.current_stream(flat.device))
            with torch.cuda.stream(side):
                self._work[bi] = self.dist.all_reduce(flat, op=self._avg_op or self.dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            self._work[bi] = self.dist.all_reduce(flat, op=self._avg_op or self.dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _on_grad(self, p):
        bi, off = self._where[p]
        if self.trace is not None:
            self.trace.append(("grad", bi))
        if not self.is_view(p, p.grad):
            self.flat[bi][off:off + p.numel()].copy_(p.grad.reshape(-1))
            self._dirty.add(p)
            self.copies_in += 1
        self._pending[bi] -= 1
        if self._pending[bi] == 0:
            self._launch(bi)

    def _settle(self, bi):
        self._work[bi].wait()         # the current stream waits for the communication stream
        if self._avg_op is None:
            self.flat[bi] /= self.world
        off = 0
        for p in self.buckets[bi]:
            if not self.is_view(p, p.grad):
                p.grad = self.flat[bi][off:off + p.numel()].view(p.shape)      # (no copy back: .grad IS the averaged slice)
            off += p.numel()
        self._pending[bi] = len(self.buckets[bi])
        self._work[bi] = None

    def finish(self):
        for bi, b in enumerate(self.buckets):
            if self._work[bi] is None:    # a parameter without gradient this step: reduce what is there (its slice as zeros)
                for p in b:
                    if p.grad is None:
                        self._slice(p).zero_()
                        self._dirty.discard(p)
                self._launch(bi)
            self._settle(bi)

    def exchange(self):
        """gradients that did not come through the hooks (a captured step replays kernels, not Python): make sure every
        gradient sits in its slice (a replayed sink-aware backward has written them there already), all-reduce, average"""
        for bi, b in enumerate(self.buckets):
            off = 0
            for p in b:
                if p.grad is None:
                    self.flat[bi][off:off + p.numel()].zero_()
                elif not self.is_view(p, p.grad):
                    self.flat[bi][off:off + p.numel()].copy_(p.grad.reshape(-1))
                    self.copies_in += 1
                off += p.numel()
            self._launch(bi)
        for bi in range(len(self.buckets)):
            self._settle(bi)
