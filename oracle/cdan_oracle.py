"""Functional CPU restatement of the reference CDAN/CBAM forward pass.

TEST INFRASTRUCTURE (see oracle/__init__.py) -- plain `torch.nn.functional`
calls on a checkpoint dictionary; no modules, no state, no product imports.
Each function cites the reference lines it follows (paths under
/root/reference/).  The arithmetic below the function level is PyTorch ATen's
CPU kernels, the same third-party code the reference runs on.

`bn_mode`:
  "eval"   running statistics (the reference's `network.eval()`, model.py:232)
  "batch"  batch statistics as in training; dropout is left out, exactly like the
           training-parity fixtures (dropout RNG note).
"""
import torch
import torch.nn.functional as F

EPS = 1e-5


def _bn(sd, p, x, bn_mode, momentum, stats_out):
    # nn.BatchNorm2d: cdan.py:12,43,50,105,109,113,116 ; cbam.py:11
    if bn_mode == "eval":
        return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"],
                            sd[p + ".weight"], sd[p + ".bias"], False, momentum, EPS)
    rm = sd[p + ".running_mean"].clone()
    rv = sd[p + ".running_var"].clone()
    y = F.batch_norm(x, rm, rv, sd[p + ".weight"], sd[p + ".bias"], True, momentum, EPS)
    if stats_out is not None:
        stats_out[p + ".running_mean"] = rm
        stats_out[p + ".running_var"] = rv
    return y


def conv_block(sd, p, x, bn_mode="eval", stats_out=None):
    """ConvBlock.forward, cdan.py:15-19: conv3x3(pad 1) -> BN -> ReLU."""
    y = F.conv2d(x, sd[p + ".conv.weight"], sd[p + ".conv.bias"], stride=1, padding=1)
    return F.relu(_bn(sd, p + ".bn", y, bn_mode, 0.1, stats_out))


def dense_block(sd, p, x, bn_mode="eval", stats_out=None, layers=4):
    """DenseBlock.forward, cdan.py:32-39 with the layer recipes at :41-53."""
    feats = x
    for i in range(layers):
        q = f"{p}.layers.{i}"
        t = F.relu(_bn(sd, q + ".0", feats, bn_mode, 0.1, stats_out))
        t = F.conv2d(t, sd[q + ".2.weight"], sd[q + ".2.bias"], stride=1, padding=1)
        feats = torch.cat((feats, t), dim=1)
    q = p + ".transition_layer"
    t = F.relu(_bn(sd, q + ".0", feats, bn_mode, 0.1, stats_out))
    return F.conv2d(t, sd[q + ".2.weight"], sd[q + ".2.bias"])


def channel_gate(sd, p, x):
    """ChannelGate.forward, cbam.py:37-60 (pool types avg + max; the MLP, and so
    its output bias, is applied to each pooled vector and the results summed)."""
    w1, b1 = sd[p + ".mlp.1.weight"], sd[p + ".mlp.1.bias"]
    w2, b2 = sd[p + ".mlp.3.weight"], sd[p + ".mlp.3.bias"]

    def mlp(v):
        return F.linear(F.relu(F.linear(v, w1, b1)), w2, b2)

    hw = (x.shape[2], x.shape[3])
    avg = F.avg_pool2d(x, hw, stride=hw).flatten(1)     # cbam.py:41
    mx = F.max_pool2d(x, hw, stride=hw).flatten(1)      # cbam.py:44 (its backward routes to ONE arg-max)
    att = mlp(avg) + mlp(mx)
    return x * torch.sigmoid(att)[:, :, None, None]


def spatial_gate(sd, p, x, bn_mode="eval", stats_out=None):
    """SpatialGate.forward, cbam.py:78-82; ChannelPool order is (max, mean),
    cbam.py:68-70; 7x7 conv without bias, BN(1) momentum 0.01, no ReLU (cbam.py:77)."""
    comp = torch.cat((torch.max(x, 1)[0].unsqueeze(1), torch.mean(x, 1).unsqueeze(1)), dim=1)   # cbam.py:68-70
    m = F.conv2d(comp, sd[p + ".spatial.conv.weight"], None, stride=1, padding=3)
    m = _bn(sd, p + ".spatial.bn", m, bn_mode, 0.01, stats_out)
    return x * torch.sigmoid(m)


def cbam(sd, p, x, bn_mode="eval", stats_out=None):
    """CBAM.forward, cbam.py:91-95."""
    return spatial_gate(sd, p + ".SpatialGate", channel_gate(sd, p + ".ChannelGate", x),
                        bn_mode, stats_out)


def up2(x):
    """F.interpolate(scale_factor=2, bilinear, align_corners=False), cdan.py:137,145,153."""
    return F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)


def encoder(sd, x, bn_mode="eval", stats_out=None):
    """Encoder.forward, cdan.py:70-98 (dropout = identity in eval)."""
    skips, denses = [], []
    t = x
    for i in (1, 2, 3):
        t = F.max_pool2d(conv_block(sd, f"encoder.conv{i}", t, bn_mode, stats_out), 2, 2)
        denses.append(dense_block(sd, f"encoder.dense{i}", t, bn_mode, stats_out))
        skips.append(t)
    e = conv_block(sd, "encoder.conv4", t, bn_mode, stats_out)
    return e, skips, denses


def _deconv_bn_relu(sd, i, x, bn_mode, stats_out):
    # ConvTranspose2d(k3, s1, p1) -> BN -> ReLU, cdan.py:127-129 (and :134-136, :142-144, :150-152)
    y = F.conv_transpose2d(x, sd[f"decoder.conv{i}.weight"], sd[f"decoder.conv{i}.bias"],
                           stride=1, padding=1)
    return F.relu(_bn(sd, f"decoder.bn{i}", y, bn_mode, 0.1, stats_out))


def decoder(sd, x, b, skips, denses, bn_mode="eval", stats_out=None, taps=None):
    """Decoder.forward, cdan.py:126-159."""
    t = _deconv_bn_relu(sd, 1, b, bn_mode, stats_out) + skips[2]
    t = cbam(sd, "decoder.cbam1", t, bn_mode, stats_out) * denses[2]
    if taps is not None:
        taps["dec1"] = t
    t = up2(_deconv_bn_relu(sd, 2, t, bn_mode, stats_out)) + skips[1]
    t = cbam(sd, "decoder.cbam2", t, bn_mode, stats_out) * denses[1]
    if taps is not None:
        taps["dec2"] = t
    t = up2(_deconv_bn_relu(sd, 3, t, bn_mode, stats_out)) + skips[0]
    t = cbam(sd, "decoder.cbam3", t, bn_mode, stats_out) * denses[0]
    if taps is not None:
        taps["dec3"] = t
    t = up2(_deconv_bn_relu(sd, 4, t, bn_mode, stats_out)) + x
    if taps is not None:
        taps["dec4"] = t
    return torch.sigmoid(dense_block(sd, "decoder.final_dense", t, bn_mode, stats_out))


def cdan_forward(sd, x, bn_mode="eval", stats_out=None, taps=None):
    """CDAN.forward, cdan.py:171-176.  `taps`, when a dict, receives the
    intermediate tensors the parity tests compare stage by stage."""
    e, skips, denses = encoder(sd, x, bn_mode, stats_out)
    b = cbam(sd, "bottleneck", e, bn_mode, stats_out)
    if taps is not None:
        taps["enc"] = e
        for i in range(3):
            taps[f"skip{i}"] = skips[i]
            taps[f"dense{i}"] = denses[i]
        taps["bott"] = b
    return decoder(sd, x, b, skips, denses, bn_mode, stats_out, taps)


def charbonnier(y, t, eps=1e-6):
    """utils/loss_factory.py:160-167: mean(sqrt((y - t)^2 + eps))."""
    return torch.sqrt((y - t) ** 2 + eps).mean()


def psnr(pred, target, data_range=1.0):
    """10*log10(range^2 / MSE) over the whole batch (torchmetrics PeakSignalNoiseRatio
    formula, utils/metrics_factory.py:76; torchmetrics itself is absent here, so
    `data_range` is passed explicitly -- "parity unpinned" at that boundary)."""
    mse = ((pred.double() - target.double()) ** 2).mean()
    return float(10.0 * torch.log10(torch.tensor(data_range ** 2, dtype=torch.float64) / mse))
