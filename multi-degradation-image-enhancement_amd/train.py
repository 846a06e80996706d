"""Training-mode CDAN (`Model.train_step`, models/model.py:138-227).

First step of the training path: every convolution -- forward, input
gradient and weight gradient, ~97 % of a step's FLOPs -- runs in libmdie_hip.so through one autograd
Function; the bandwidth-bound glue between them (batch-statistic BatchNorm, ReLU, max-pool, dropout,
CBAM gates, bilinear upsampling, the loss) is composed from PyTorch-ROCm device ops and differentiated
by autograd.  Nothing here touches the CPU or the oracle.  Tensors are `channels_last`, which IS the
engine's NHWC layout, so the two worlds share buffers without copies.

Reference semantics kept: BatchNorm uses batch statistics and updates running statistics with momentum
0.1 (0.01 in CBAM's spatial gate, models/cbam.py:11), eps 1e-5; dropout p=0.2 after each encoder stage
with the dense blocks fed from the PRE-dropout tensor (models/cdan.py:76-79); ConvTranspose2d weights
stay in their [Cin, Cout, kh, kw] checkpoint layout.
"""
import ctypes as C

import torch
import torch.nn.functional as F

from . import engine as E
from . import lib as L

EPS = 1e-5


def _cl(t):
    return t.contiguous(memory_format=torch.channels_last)


def _nhwc(t):
    """[B,C,H,W] channels_last tensor -> (data_ptr, C, pixel stride in elements)."""
    assert t.is_contiguous(memory_format=torch.channels_last) or t.shape[1] == 1
    return t.data_ptr(), t.shape[1], t.stride(3)


def _pack(dtype, w, ks, transposed, cout, cin):
    n = L.lib.mdie_conv_weight_bytes(dtype, ks, cin, cout)
    dst = torch.empty(n, dtype=torch.uint8, device=w.device)
    L.check(L.lib.mdie_pack_conv_weight_dev(dtype, ks, int(transposed), w.data_ptr(), cout, cin, cout, cin, cin, 0, dst.data_ptr(),
                                            E._stream_ptr(w.device)), "mdie_pack_conv_weight_dev")
    return dst


def _launch_conv(dtype, segs, packed, bias_f32, ks, cout, out):
    B, _, H, W = segs[0].shape
    d = L.ConvDesc()
    d.dtype, d.B, d.H, d.W, d.ksize = dtype, B, H, W, ks
    d.nseg = len(segs)
    cin = 0
    for i, s in enumerate(segs):
        p, c, st = _nhwc(s)
        d.inp[i] = L.Seg(p, c, st)
        cin += c
    d.cin, d.cout = cin, cout
    d.pre_scale = d.pre_shift = None
    ones = torch.ones(cout, dtype=torch.float32, device=out.device)
    d.weight, d.post_scale, d.post_shift = packed.data_ptr(), ones.data_ptr(), bias_f32.data_ptr()
    d.act, d.pool = L.ACT_NONE, 0
    d.residual, d.res_stride = None, 0
    d.out, d.out_stride = out.data_ptr(), out.stride(3)
    d.out_nchw3 = None
    L.check(L.lib.mdie_conv_fwd(C.byref(d), E._stream_ptr(out.device)), "mdie_conv_fwd")
    return ones  # keep alive until the launch is enqueued on this stream (same-stream ordering)


class _ConvFn(torch.autograd.Function):
    """y = conv_k(cat(segments), weight) + bias on the HIP engine; stored channel counts are multiples of 16.
    weight: [cout, cin, k, k] (transposed=False) or [cin, cout, k, k] (transposed=True, ConvTranspose2d k3 s1 p1)."""

    @staticmethod
    def forward(ctx, dtype, transposed, weight, bias, *segs):
        ks = weight.shape[2]
        cout, cin = (weight.shape[1], weight.shape[0]) if transposed else (weight.shape[0], weight.shape[1])
        assert cin == sum(s.shape[1] for s in segs) and cin % 16 == 0 and cout % 16 == 0
        w32 = weight.detach().float().contiguous()
        packed = _pack(dtype, w32, ks, transposed, cout, cin)
        B, _, H, W = segs[0].shape
        out = torch.empty(B, cout, H, W, dtype=E.TORCH_DTYPE[dtype], device=weight.device, memory_format=torch.channels_last)
        _launch_conv(dtype, segs, packed, bias.detach().float().contiguous(), ks, cout, out)
        ctx.save_for_backward(w32, *segs)
        ctx.meta = (dtype, transposed, ks, cout, cin)
        return out

    @staticmethod
    def backward(ctx, dy):
        dtype, transposed, ks, cout, cin = ctx.meta
        w32, *segs = ctx.saved_tensors
        dy = dy.to(E.TORCH_DTYPE[dtype]).contiguous(memory_format=torch.channels_last)
        B, _, H, W = dy.shape
        dev = dy.device
        # ---- input gradient: the same convolution with the in/out-swapped, flipped kernel ----------------
        packed = _pack(dtype, w32, ks, not transposed, cin, cout)
        dx = torch.empty(B, cin, H, W, dtype=dy.dtype, device=dev, memory_format=torch.channels_last)
        zero_bias = torch.zeros(cin, dtype=torch.float32, device=dev)
        _launch_conv(dtype, [dy], packed, zero_bias, ks, cin, dx)
        # ---- weight gradient -----------------------------------------------------------------------------------
        dw = torch.empty_like(w32)
        nws = L.lib.mdie_conv_wgrad_workspace_bytes(B, H, W, ks, cin, cout)
        ws = torch.empty(nws, dtype=torch.uint8, device=dev)
        d = L.WgradDesc()
        d.dtype, d.B, d.H, d.W, d.ksize, d.transposed = dtype, B, H, W, ks, int(transposed)
        d.nseg = len(segs)
        for i, s in enumerate(segs):
            p, c, st = _nhwc(s)
            d.inp[i] = L.Seg(p, c, st)
        d.cin, d.cout, d.cout_stored, d.split, d.gap = cin, cout, cout, cin, 0
        d.dy, d.dy_stride = dy.data_ptr(), dy.stride(3)
        d.dw, d.workspace, d.workspace_bytes = dw.data_ptr(), ws.data_ptr(), nws
        L.check(L.lib.mdie_conv_wgrad(C.byref(d), E._stream_ptr(dev)), "mdie_conv_wgrad")
        db = dy.float().sum(dim=(0, 2, 3))
        grads, c0 = [], 0
        for s in segs:
            grads.append(dx[:, c0:c0 + s.shape[1]])
            c0 += s.shape[1]
        return (None, None, dw, db, *grads)


def conv(dtype, weight, bias, segs, transposed=False):
    return _ConvFn.apply(dtype, transposed, weight, bias, *[_cl(s) for s in segs])


def _pad_c(t, c):
    return t if t.shape[1] == c else _cl(F.pad(t, (0, 0, 0, 0, 0, c - t.shape[1])))


def _bn(node, x, momentum, lo=None, hi=None):
    """Batch-statistic BatchNorm over channels [lo:hi] of `node`'s parameters (per-channel, so a BN over a
    concatenation is the BN of each segment with the matching parameter slice)."""
    sl = slice(lo, hi)
    return F.batch_norm(x, node.running_mean[sl], node.running_var[sl], node.weight[sl], node.bias[sl], True, momentum, EPS)


def _tick(net):
    for name, buf in net.named_buffers():
        if name.endswith("num_batches_tracked"):
            buf += 1


def _dense_block(net_dtype, blk, x, real_c):
    """DenseBlock.forward (models/cdan.py:32-39): x has `real_c` real channels (3 for final_dense, stored in a
    tensor of its own); growth segments are 16 channels each."""
    layers = blk.layers
    segs = [x]
    for i in range(4):
        bn, cv = getattr(layers, str(i))._modules["0"], getattr(layers, str(i))._modules["2"]
        acts, c0 = [], 0
        for j, s in enumerate(segs):
            w = s.shape[1]
            a = F.relu(_bn(bn, s, 0.1, c0, c0 + w))
            acts.append(_pad_c(a, 16) if w < 16 else a)
            c0 += w
        weight = cv.weight
        if real_c < 16:  # place the 3 real base channels inside their 16 stored ones (zero weights for the padding)
            weight = torch.cat((weight[:, :real_c], weight.new_zeros(weight.shape[0], 16 - real_c, 3, 3), weight[:, real_c:]), 1)
        segs.append(conv(net_dtype, weight, cv.bias, acts))
    bn, cv = blk.transition_layer._modules["0"], blk.transition_layer._modules["2"]
    acts, c0 = [], 0
    for s in segs:
        w = s.shape[1]
        a = F.relu(_bn(bn, s, 0.1, c0, c0 + w))
        acts.append(_pad_c(a, 16) if w < 16 else a)
        c0 += w
    weight, bias = cv.weight, cv.bias
    if real_c < 16:
        weight = torch.cat((weight[:, :real_c], weight.new_zeros(weight.shape[0], 16 - real_c, 1, 1), weight[:, real_c:]), 1)
        weight = F.pad(weight, (0, 0, 0, 0, 0, 0, 0, 16 - weight.shape[0]))
        bias = F.pad(bias, (0, 16 - bias.shape[0]))
    y = conv(net_dtype, weight, bias, acts)
    return y[:, :real_c] if real_c < 16 else y


def _cbam(node, x):
    """CBAM.forward in training mode (models/cbam.py:37-60, 68-82, 91-95)."""
    mlp = node.ChannelGate.mlp
    w1, b1, w2, b2 = mlp._modules["1"].weight, mlp._modules["1"].bias, mlp._modules["3"].weight, mlp._modules["3"].bias

    def gate(v):
        return F.linear(F.relu(F.linear(v, w1.to(v.dtype), b1.to(v.dtype))), w2.to(v.dtype), b2.to(v.dtype))

    att = gate(x.mean(dim=(2, 3))) + gate(x.amax(dim=(2, 3)))
    xg = x * torch.sigmoid(att)[:, :, None, None]
    sp = node.SpatialGate.spatial
    comp = torch.stack((xg.amax(dim=1), xg.mean(dim=1)), dim=1)
    m = F.conv2d(comp.float(), sp.conv.weight, None, padding=3)
    m = F.batch_norm(m, sp.bn.running_mean, sp.bn.running_var, sp.bn.weight, sp.bn.bias, True, 0.01, EPS)
    return xg * torch.sigmoid(m).to(xg.dtype)


def _up2(t):
    return F.interpolate(t, scale_factor=2, mode="bilinear", align_corners=False)


def forward_train(net, x, precision="fp32", dropout_p=0.2):
    """CDAN.forward (models/cdan.py:171-176) with the module in training mode.  x: fp32 NCHW on the GPU."""
    if not x.is_cuda:
        raise L.MdieError("forward_train: GPU tensors only (no CPU fallback)")
    dt = E.dtype_id(precision)
    td = E.TORCH_DTYPE[dt]
    enc, dec = net.encoder, net.decoder
    _tick(net)

    def drop(t):
        return F.dropout(t, dropout_p, True) if dropout_p > 0 else t

    xin = _cl(x.to(td))
    t = _pad_c(xin, 16)
    skips, denses = [], []
    for i in (1, 2, 3):
        blk = getattr(enc, f"conv{i}")
        w = blk.conv.weight
        if i == 1:
            w = F.pad(w, (0, 0, 0, 0, 0, 13))  # 3 real input channels inside 16 stored ones
        y = F.relu(_bn(blk.bn, conv(dt, w, blk.conv.bias, [t]), 0.1))
        o = _cl(F.max_pool2d(y, 2, 2))
        denses.append(_dense_block(dt, getattr(enc, f"dense{i}"), o, o.shape[1]))
        t = _cl(drop(o))
        skips.append(t)
    e = _cl(drop(F.relu(_bn(enc.conv4.bn, conv(dt, enc.conv4.conv.weight, enc.conv4.conv.bias, [t]), 0.1))))
    t = _cl(_cbam(net.bottleneck, e))

    def deconv(i, inp):
        cv, bn = getattr(dec, f"conv{i}"), getattr(dec, f"bn{i}")
        w, b = cv.weight, cv.bias
        if w.shape[1] < 16:  # decoder.conv4: 3 real output channels in 16 stored ones
            w, b = F.pad(w, (0, 0, 0, 0, 0, 16 - w.shape[1])), F.pad(b, (0, 16 - b.shape[0]))
        y = conv(dt, w, b, [inp], transposed=True)
        y = y[:, :cv.weight.shape[1]]
        return F.relu(_bn(bn, y, 0.1))

    t = deconv(1, t) + skips[2]
    t = _cl(_cbam(dec.cbam1, t) * denses[2])
    t = _up2(deconv(2, t)) + skips[1]
    t = _cl(_cbam(dec.cbam2, _cl(t)) * denses[1])
    t = _up2(deconv(3, t)) + skips[0]
    t = _cl(_cbam(dec.cbam3, _cl(t)) * denses[0])
    t = _cl(_up2(deconv(4, t)) + xin)
    return torch.sigmoid(_dense_block(dt, dec.final_dense, t, 3)).float().contiguous()


# ---- data-parallel gradient exchange ---------------------------------------------------------------------
class GradBuckets:
    """Bucketed gradient all-reduce overlapped with backward.

    Parameters are grouped in REVERSE registration order (gradients become ready roughly back-to-front)
    into `n_buckets` flat fp32 buckets of about equal size (the whole model is 14.3 MB: PyTorch-DDP's 25 MB
    default would be ONE bucket and zero overlap).  A post-accumulate-grad hook copies each gradient into
    its bucket; when the last gradient of a bucket has arrived the bucket's all-reduce is launched
    asynchronously (RCCL over xGMI with the "nccl" backend), so it runs under the rest of backward.
    `finish()` waits, divides by the world size and scatters the averages back into `.grad`."""

    def __init__(self, params, process_group=None, n_buckets=4):
        import torch.distributed as dist
        self.dist, self.group = dist, process_group
        self.world = dist.get_world_size(process_group)
        params = [p for p in params if p.requires_grad][::-1]
        total = sum(p.numel() for p in params)
        target = max(1, -(-total // n_buckets))
        self.buckets, cur, size = [], [], 0
        for p in params:
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
        self._pending = [len(b) for b in self.buckets]
        self._work = [None] * len(self.buckets)
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for b in self.buckets for p in b]

    def _on_grad(self, p):
        bi, off = self._where[p]
        self.flat[bi][off:off + p.numel()].copy_(p.grad.reshape(-1))
        self._pending[bi] -= 1
        if self._pending[bi] == 0:
            self._work[bi] = self.dist.all_reduce(self.flat[bi], op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish(self):
        for bi, b in enumerate(self.buckets):
            if self._pending[bi] != 0:  # a parameter without gradient this step: reduce what is there
                for p in b:
                    if p.grad is None:
                        _, off = self._where[p]
                        self.flat[bi][off:off + p.numel()].zero_()
                self._work[bi] = self.dist.all_reduce(self.flat[bi], op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._work[bi].wait()
            self.flat[bi] /= self.world
            off = 0
            for p in b:
                g = self.flat[bi][off:off + p.numel()].view_as(p)
                if p.grad is None:
                    p.grad = g.clone()
                else:
                    p.grad.copy_(g)
                off += p.numel()
            self._pending[bi] = len(b)
            self._work[bi] = None

    def remove(self):
        for h in self._hooks:
            h.remove()
