"""Degradation classifier / router (BASELINE configs[3]).

The reference trains a multi-label classifier (`classification/train_multilabel_classifier.py:117-131`: torchvision
ResNet18 backbone, `fc = Identity`, heads `head_cls` / `head_sev` = nn.Linear(512, num_classes)), evaluates it on
images resized to 256x384 and ImageNet-normalised (:760-776), tunes one threshold per class (:251-304) and saves
`{"model_state", "classes", "normalize", "imagenet_mean", "imagenet_std", "default_thresh"}` (:860-871).  It ships no
code that USES the classifier to pick an enhancer; the routing policy below is this repository's:

    probs = sigmoid(cls_logits);  detected = {c : probs[c] >= threshold[c]}
    task  = the detected class with the largest margin probs[c] - threshold[c], or None (clean image: pass through)

Inference runs on the HIP engine: 7x7/s2 stem (`mdie_stem7_fwd`), 3x3/s2 max-pool, sixteen 3x3 and three 1x1
convolutions on `mdie_conv_fwd` (eval BatchNorm folded, identity branch as the epilogue's residual followed by
`mdie_relu_inplace`, stride 2 = stride 1 + `mdie_subsample2`), global average pool + both heads + sigmoid in `mdie_avgpool_heads`.
GPU only, no fallback.  ImageNet weights cannot be downloaded here: tests use seeded random parameters.
"""
import json
from collections import OrderedDict

import numpy as np
import torch

from . import engine as E
from . import lib as L

CLASSES = ("blur", "noise", "low_light", "jpeg", "pixelation", "motion_blur", "high_light", "low_contrast", "color_distortion")
IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
_STAGES = ((64, 1), (128, 2), (256, 2), (512, 2))     # resnet18: two BasicBlocks per stage
EPS = 1e-5


def router_param_spec(num_classes=len(CLASSES)):
    """state_dict keys and shapes of MultiHeadClassifier (torchvision.models.resnet18 under `backbone.`)."""
    spec = OrderedDict()

    def bn(p, c):
        spec[p + ".weight"], spec[p + ".bias"] = (c,), (c,)
        spec[p + ".running_mean"], spec[p + ".running_var"], spec[p + ".num_batches_tracked"] = (c,), (c,), ()

    spec["backbone.conv1.weight"] = (64, 3, 7, 7)
    bn("backbone.bn1", 64)
    cin = 64
    for li, (c, stride) in enumerate(_STAGES, start=1):
        for b in range(2):
            p = f"backbone.layer{li}.{b}"
            spec[p + ".conv1.weight"] = (c, cin if b == 0 else c, 3, 3)
            bn(p + ".bn1", c)
            spec[p + ".conv2.weight"] = (c, c, 3, 3)
            bn(p + ".bn2", c)
            if b == 0 and (stride != 1 or cin != c):
                spec[p + ".downsample.0.weight"] = (c, cin, 1, 1)
                bn(p + ".downsample.1", c)
        cin = c
    for h in ("head_cls", "head_sev"):
        spec[h + ".weight"], spec[h + ".bias"] = (num_classes, 512), (num_classes,)
    return spec


class DegradationRouter:
    def __init__(self, device="cuda", precision="bf16"):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise L.MdieError(f"DegradationRouter needs a GPU device, got {self.device} (no CPU fallback)")
        self.dtype = E.dtype_id(precision)
        self.classes = list(CLASSES)
        self.thresholds = None
        self.normalize = True
        self.mean, self.std = IMAGENET_MEAN, IMAGENET_STD
        self.default_thresh = 0.5
        self._p = None

    # ---- parameters -------------------------------------------------------------------------------------------------
    def load(self, checkpoint):
        """`checkpoint`: the dict the reference saves (:860-871) or a bare MultiHeadClassifier state_dict."""
        sd = checkpoint
        if isinstance(checkpoint, dict) and "model_state" in checkpoint:
            sd = checkpoint["model_state"]
            self.classes = list(checkpoint.get("classes", self.classes))
            self.normalize = bool(checkpoint.get("normalize", True))
            self.mean = tuple(checkpoint.get("imagenet_mean", IMAGENET_MEAN))
            self.std = tuple(checkpoint.get("imagenet_std", IMAGENET_STD))
            self.default_thresh = float(checkpoint.get("default_thresh", 0.5))
        spec = router_param_spec(len(self.classes))
        missing = [k for k in spec if k not in sd]
        if missing:
            raise L.MdieError(f"DegradationRouter.load: state_dict lacks {missing[:4]}{'...' if len(missing) > 4 else ''}")
        for k, shape in spec.items():
            if tuple(sd[k].shape) != tuple(shape):
                raise L.MdieError(f"DegradationRouter.load: {k} has shape {tuple(sd[k].shape)}, expected {tuple(shape)}")
        f = lambda k: sd[k].detach().to("cpu", torch.float32)
        dev = self.device

        def fold(p):                       # eval BatchNorm -> per-channel scale / shift after the convolution
            s = f(p + ".weight") / torch.sqrt(f(p + ".running_var") + EPS)
            return s.to(dev), (f(p + ".bias") - f(p + ".running_mean") * s).to(dev)

        P = {}
        w = np.ascontiguousarray(f("backbone.conv1.weight").numpy())
        blob = torch.zeros(L.lib.mdie_stem7_weight_bytes(self.dtype), dtype=torch.uint8)
        L.check(L.lib.mdie_pack_stem7_weight(self.dtype, w.ctypes.data, blob.data_ptr()), "mdie_pack_stem7_weight")
        P["stem"] = (blob.to(dev),) + fold("backbone.bn1")
        for li, (c, stride) in enumerate(_STAGES, start=1):
            for b in range(2):
                p = f"backbone.layer{li}.{b}"
                blk = {"c": c, "stride": stride if b == 0 else 1,
                       "conv1": (E.pack_conv_weight(f(p + ".conv1.weight"), self.dtype).to(dev),) + fold(p + ".bn1"),
                       "conv2": (E.pack_conv_weight(f(p + ".conv2.weight"), self.dtype).to(dev),) + fold(p + ".bn2")}
                if p + ".downsample.0.weight" in spec:
                    blk["down"] = (E.pack_conv_weight(f(p + ".downsample.0.weight"), self.dtype).to(dev),) + fold(p + ".downsample.1")
                P[p] = blk
        P["heads"] = tuple(f(k).contiguous().to(dev) for k in ("head_cls.weight", "head_cls.bias", "head_sev.weight", "head_sev.bias"))
        self._p = P
        return self

    def load_thresholds(self, thresholds):
        """A tuning report (`{"thresholds": {class: value}}`, :295-304), its path, a {class: value} dict or a list."""
        if isinstance(thresholds, str):
            with open(thresholds) as fh:
                thresholds = json.load(fh)
        if isinstance(thresholds, dict):
            thresholds = thresholds.get("thresholds", thresholds)
            thresholds = [float(thresholds.get(c, self.default_thresh)) for c in self.classes]
        if len(thresholds) != len(self.classes):
            raise L.MdieError(f"DegradationRouter.load_thresholds: {len(thresholds)} thresholds for {len(self.classes)} classes")
        self.thresholds = [float(t) for t in thresholds]
        return self

    # ---- inference -----------------------------------------------------------------------------------------------------
    def _conv(self, x, params, ks, cout, act, residual=None):
        """act(conv * scale + shift), or relu(conv * scale + shift + residual) when a residual is given"""
        w, s, b = params
        if residual is None:
            return E.conv_fwd([x], w, s, b, dtype=self.dtype, ksize=ks, cout=cout, act=act)
        y = E.conv_fwd([x], w, s, b, dtype=self.dtype, ksize=ks, cout=cout, act=L.ACT_NONE, residual=residual)
        B, H, W, Cc = y.shape
        L.check(L.lib.mdie_relu_inplace(self.dtype, B * H * W, Cc, y.data_ptr(), Cc, E._stream_ptr(y.device)), "mdie_relu_inplace")
        return y

    def _sub2(self, x):
        B, H, W, Cc = x.shape
        out = torch.empty(B, (H + 1) // 2, (W + 1) // 2, Cc, dtype=x.dtype, device=x.device)
        L.check(L.lib.mdie_subsample2(self.dtype, B, H, W, Cc, x.data_ptr(), x.stride(2), out.data_ptr(), Cc, E._stream_ptr(x.device)), "mdie_subsample2")
        return out

    def forward(self, x):
        """x: float [B,3,H,W] in [0,1] on the GPU (the reference feeds 256x384).  -> (prob_cls [B,nc], severity [B,nc]) fp32"""
        if self._p is None:
            raise L.MdieError("DegradationRouter.forward before load()")
        E._require_gpu(x, "DegradationRouter.forward")
        if x.dim() != 4 or x.shape[1] != 3:
            raise L.MdieError(f"DegradationRouter.forward: expected [B,3,H,W], got {tuple(x.shape)}")
        x = x.to(torch.float32).contiguous()
        B, _, H, W = x.shape
        dev, td, sp = x.device, E.TORCH_DTYPE[self.dtype], E._stream_ptr(x.device)
        import ctypes as C
        mean = (C.c_float * 3)(*self.mean) if self.normalize else None
        std = (C.c_float * 3)(*self.std) if self.normalize else None
        w, s, b = self._p["stem"]
        h1, w1 = (H + 1) // 2, (W + 1) // 2
        t = torch.empty(B, h1, w1, 64, dtype=td, device=dev)
        L.check(L.lib.mdie_stem7_fwd(self.dtype, B, H, W, x.data_ptr(), mean, std, w.data_ptr(), s.data_ptr(), b.data_ptr(), t.data_ptr(), 64, sp),
                "mdie_stem7_fwd")
        h2, w2 = (h1 + 1) // 2, (w1 + 1) // 2
        y = torch.empty(B, h2, w2, 64, dtype=td, device=dev)
        L.check(L.lib.mdie_maxpool3x3s2(self.dtype, B, h1, w1, 64, t.data_ptr(), 64, y.data_ptr(), 64, sp), "mdie_maxpool3x3s2")
        for li in range(1, 5):
            for bi in range(2):
                blk = self._p[f"backbone.layer{li}.{bi}"]
                c = blk["c"]
                t = self._conv(y, blk["conv1"], 3, c, L.ACT_RELU)
                idn = y
                if blk["stride"] == 2:
                    t = self._sub2(t)          # conv3x3 stride 2 == stride 1 sampled at even pixels
                    idn = self._sub2(y)
                if "down" in blk:
                    idn = self._conv(idn, blk["down"], 1, c, L.ACT_NONE)
                y = self._conv(t, blk["conv2"], 3, c, L.ACT_RELU, residual=idn)
        wc, bc, ws, bs = self._p["heads"]
        nc = len(self.classes)
        probs = torch.empty(B, nc, dtype=torch.float32, device=dev)
        sev = torch.empty(B, nc, dtype=torch.float32, device=dev)
        _, fh, fw, fc = y.shape
        L.check(L.lib.mdie_avgpool_heads(self.dtype, B, fh, fw, fc, y.data_ptr(), fc, wc.data_ptr(), bc.data_ptr(), ws.data_ptr(), bs.data_ptr(), nc,
                                         None, probs.data_ptr(), sev.data_ptr(), sp), "mdie_avgpool_heads")
        return probs, sev

    def route(self, x):
        """-> (labels, probs): per image the task to enhance with (a class name) or None for "looks clean"."""
        probs, _ = self.forward(x)
        thr = torch.tensor(self.thresholds if self.thresholds is not None else [self.default_thresh] * len(self.classes), device=probs.device)
        margin = probs - thr
        best = margin.argmax(dim=1)
        hit = margin.gather(1, best[:, None])[:, 0] >= 0
        best, hit = best.cpu().tolist(), hit.cpu().tolist()          # one small D2H per batch
        return [self.classes[i] if h else None for i, h in zip(best, hit)], probs
