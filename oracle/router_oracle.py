"""CPU restatement of the degradation classifier's forward (classification/train_multilabel_classifier.py:117-131).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  The reference builds `torchvision.models.resnet18` -- torchvision is
not installed in this image and the reference holds no test or fixture for the classifier, so the backbone is restated
from torchvision's published definition (conv7x7/s2/p3 -> BN -> ReLU -> maxpool3x3/s2/p1 -> 4 stages of two BasicBlocks
[conv3x3(s) -> BN -> ReLU -> conv3x3 -> BN, + identity or conv1x1(s)+BN downsample, ReLU] -> adaptive avg-pool -> flatten):
PARITY UNPINNED at the torchvision boundary.  Heads, normalisation constants (:760) and the sigmoid are the reference's.
"""
import torch
import torch.nn.functional as F

EPS = 1e-5
STAGES = ((64, 1), (128, 2), (256, 2), (512, 2))


def _bn(sd, p, x):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.1, EPS)


def basic_block(sd, p, x, stride):
    out = F.relu(_bn(sd, p + ".bn1", F.conv2d(x, sd[p + ".conv1.weight"], None, stride=stride, padding=1)))
    out = _bn(sd, p + ".bn2", F.conv2d(out, sd[p + ".conv2.weight"], None, stride=1, padding=1))
    idn = x
    if p + ".downsample.0.weight" in sd:
        idn = _bn(sd, p + ".downsample.1", F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride=stride))
    return F.relu(out + idn)


def backbone(sd, x):
    x = F.relu(_bn(sd, "backbone.bn1", F.conv2d(x, sd["backbone.conv1.weight"], None, stride=2, padding=3)))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    for li, (_, stride) in enumerate(STAGES, start=1):
        for b in range(2):
            x = basic_block(sd, f"backbone.layer{li}.{b}", x, stride if b == 0 else 1)
    return torch.flatten(F.adaptive_avg_pool2d(x, 1), 1)


def classifier_forward(sd, x, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225), normalize=True):
    """x in [0,1] -> (sigmoid(cls_logits), sigmoid(sev_logits)); the evaluation loop applies both sigmoids (:235-236)."""
    if normalize:
        x = (x - torch.tensor(mean, dtype=x.dtype).view(1, 3, 1, 1)) / torch.tensor(std, dtype=x.dtype).view(1, 3, 1, 1)
    feat = backbone(sd, x)
    cls = F.linear(feat, sd["head_cls.weight"], sd["head_cls.bias"])
    sev = F.linear(feat, sd["head_sev.weight"], sd["head_sev.bias"])
    return torch.sigmoid(cls), torch.sigmoid(sev)


def route(probs, thresholds, classes):
    """the repository's routing policy (mdie_amd.router): largest margin over its threshold, else None"""
    margin = probs - torch.tensor(thresholds, dtype=probs.dtype)
    out = []
    for m in margin:
        i = int(m.argmax())
        out.append(classes[i] if float(m[i]) >= 0 else None)
    return out
