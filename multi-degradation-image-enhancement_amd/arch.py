"""Checkpoint layout of the CDAN network as data.

The drop-in contract is the key set of `CDAN().state_dict()`:
236 entries whose names follow the reference's module attributes
(/root/reference/models/cdan.py:55-176, /root/reference/models/cbam.py:26-95).
The table below generates those names from the layer widths; nothing here is
executable network code -- the network itself runs in libmdie_hip.so.
"""
from collections import OrderedDict

ENCODER_WIDTHS = (3, 64, 128, 256, 512)      # encoder.conv1..4 (cdan.py:58-61)
DECODER_WIDTHS = (512, 256, 128, 64, 3)      # decoder.conv1..4 (cdan.py:103-115)
GROWTH, DENSE_LAYERS, REDUCTION = 16, 4, 16  # DenseBlock(c, c, 16, 4) (cdan.py:63-65,119); CBAM reduction (cbam.py:27)

_BN_LEAVES = (("weight", "param"), ("bias", "param"), ("running_mean", "buffer"),
              ("running_var", "buffer"), ("num_batches_tracked", "counter"))


def _batchnorm(prefix, channels):
    for leaf, kind in _BN_LEAVES:
        yield f"{prefix}.{leaf}", (() if kind == "counter" else (channels,)), kind


def _dense(prefix, channels):
    width = channels
    for layer in range(DENSE_LAYERS):
        yield from _batchnorm(f"{prefix}.layers.{layer}.0", width)
        yield f"{prefix}.layers.{layer}.2.weight", (GROWTH, width, 3, 3), "param"
        yield f"{prefix}.layers.{layer}.2.bias", (GROWTH,), "param"
        width += GROWTH
    yield from _batchnorm(f"{prefix}.transition_layer.0", width)
    yield f"{prefix}.transition_layer.2.weight", (channels, width, 1, 1), "param"
    yield f"{prefix}.transition_layer.2.bias", (channels,), "param"


def _attention(prefix, channels):
    hidden = channels // REDUCTION
    yield f"{prefix}.ChannelGate.mlp.1.weight", (hidden, channels), "param"
    yield f"{prefix}.ChannelGate.mlp.1.bias", (hidden,), "param"
    yield f"{prefix}.ChannelGate.mlp.3.weight", (channels, hidden), "param"
    yield f"{prefix}.ChannelGate.mlp.3.bias", (channels,), "param"
    yield f"{prefix}.SpatialGate.spatial.conv.weight", (1, 2, 7, 7), "param"
    yield from _batchnorm(f"{prefix}.SpatialGate.spatial.bn", 1)


def _entries():
    for i in range(4):
        cin, cout = ENCODER_WIDTHS[i], ENCODER_WIDTHS[i + 1]
        yield f"encoder.conv{i + 1}.conv.weight", (cout, cin, 3, 3), "param"
        yield f"encoder.conv{i + 1}.conv.bias", (cout,), "param"
        yield from _batchnorm(f"encoder.conv{i + 1}.bn", cout)
    for i in range(3):
        yield from _dense(f"encoder.dense{i + 1}", ENCODER_WIDTHS[i + 1])
    yield from _attention("bottleneck", ENCODER_WIDTHS[4])
    for i in range(4):
        cin, cout = DECODER_WIDTHS[i], DECODER_WIDTHS[i + 1]
        yield f"decoder.conv{i + 1}.weight", (cin, cout, 3, 3), "param"  # ConvTranspose2d layout
        yield f"decoder.conv{i + 1}.bias", (cout,), "param"
        if i < 3:
            yield from _attention(f"decoder.cbam{i + 1}", cout)
        yield from _batchnorm(f"decoder.bn{i + 1}", cout)
    yield from _dense("decoder.final_dense", DECODER_WIDTHS[4])


def cdan_param_spec():
    """OrderedDict name -> (shape, kind) with kind in {"param", "buffer", "counter"},
    in the order `CDAN().state_dict()` lists them."""
    return OrderedDict((name, (shape, kind)) for name, shape, kind in _entries())


def cbam_param_spec(channels):
    return OrderedDict((name[2:], (shape, kind)) for name, shape, kind in _attention("m", channels))
