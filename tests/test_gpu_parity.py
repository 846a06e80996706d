"""Parity of the HIP path (through the C ABI of libmdie_hip.so) against

  * the golden vectors the reference itself produced (tests/golden/*.npz), and
  * the oracle (oracle/cdan_oracle.py, pinned by those vectors) on fresh seeded inputs.

Tolerances (north star: "within 1e-3 relative fp32"), every bound set from the values measured on MI355X
(MDIE_ERRLOG=file python -m pytest tests -m gpu logs each one; round-2 log: profiles/r02a_parity_errors.txt):
  fp32 path : max|hip - ref| / max|ref| <= 1e-3 is the contract; the kernels are asserted
              at 2e-5 (exact-f32 MFMA, only summation order differs; measured <= 2.2e-6).
  fp16 path : fp16 storage, fp32 accumulation (the reference's own autocast dtype): network outputs are held to the
              1e-3 CONTRACT itself (measured 2.4e-4 ... 7.9e-4), single operators and stage taps to 1.4e-3.
  bf16 path : bf16 storage with fp32 accumulation cannot meet 1e-3 (the reference under bf16
              autocast is itself 4.2e-3 off): outputs and operators <= 8e-3 (measured
              2.2e-3 ... 5.4e-3), stage taps deep in the network <= 1.2e-2 (measured <= 7.0e-3), and >= 46 dB PSNR
              against the fp32 reference output (measured 61-65 dB).
"""
import math
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

FP32_TOL = 2e-5
CONTRACT_TOL = 1e-3
BF16_TOL = 8e-3         # outputs / single operators (measured <= 5.4e-3)
BF16_TAP_TOL = 1.2e-2   # intermediate taps of the whole network (measured <= 7.0e-3)
F16_TOL = 1.4e-3        # single operators and taps (measured <= 7.9e-4)
F16_OUT_TOL = CONTRACT_TOL   # network outputs in fp16 meet the north-star bound itself
PRECISIONS = ["fp32", "bf16", "fp16"]
TORCH_DT = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}
_ERRLOG = os.environ.get("MDIE_ERRLOG")   # measurement aid: every rel_to_max value with its test id (how the bounds above were set)


@pytest.fixture(scope="module")
def E():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import mdie_amd.engine as eng
    return eng


@pytest.fixture(scope="module")
def L():
    import mdie_amd.lib as lib
    return lib


def _golden(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name))
    arrays = {k: torch.from_numpy(z[k]) for k in z.files if not k.startswith("p:") and z[k].dtype == np.float32}
    params = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("p:")}
    return arrays, params


def rel_to_max(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    v = ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()
    if _ERRLOG:
        with open(_ERRLOG, "a") as f:
            f.write(f"{os.environ.get('PYTEST_CURRENT_TEST', '?')}\t{v:.4e}\n")
    return v


def psnr(a, b):
    mse = ((a.double().cpu() - b.double().cpu()) ** 2).mean().item()
    return 10 * math.log10(1.0 / max(mse, 1e-20))


def tol_for(precision):
    return {"fp32": FP32_TOL, "bf16": BF16_TOL, "fp16": F16_TOL}[precision]


def out_tol_for(precision):
    """bound on the network's final output"""
    return {"fp32": FP32_TOL, "bf16": BF16_TOL, "fp16": F16_OUT_TOL}[precision]


def bn_fold(p, prefix):
    s = p[prefix + ".weight"] / torch.sqrt(p[prefix + ".running_var"] + 1e-5)
    return s, p[prefix + ".bias"] - p[prefix + ".running_mean"] * s


def pad_c(x, c):
    if x.shape[1] == c:
        return x
    return torch.cat((x, x.new_zeros(x.shape[0], c - x.shape[1], *x.shape[2:])), 1)


def pad_v(v, n):
    out = torch.zeros(n)
    out[:v.numel()] = v
    return out


# ---------------------------------------------------------------------------------------------------------------------
# whole network
# ---------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def net():
    from models.cdan import CDAN
    from oracle import params as P
    m = CDAN()
    m.load_state_dict(P.make_state_dict(42), strict=True)
    return m.eval().cuda()


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("tag", ["1x32x32", "1x40x56", "2x64x64_lowlight", "4x64x64_noise"])
def test_e2e_golden(E, net, golden_dir, tag, precision):
    g, _ = _golden(golden_dir, f"e2e_eval_{tag}.npz")
    net.precision = precision
    with torch.no_grad():
        y, taps = net.forward_with_taps(g["x"].cuda())
    tap_tol = {"fp32": FP32_TOL, "bf16": BF16_TAP_TOL, "fp16": F16_TOL}[precision]
    for k in ("skip0", "dense0", "skip1", "dense1", "skip2", "dense2", "enc", "bott"):
        if k in g:
            err = rel_to_max(taps[k], g[k])
            assert err <= tap_tol, f"{k}: {err:.3e}"
    err = rel_to_max(y, g["y"])
    print(f"[{precision}] {tag}: rel-to-max {err:.3e}  PSNR vs reference {psnr(y, g['y']):.1f} dB")
    assert err <= out_tol_for(precision)
    if precision != "bf16":
        assert err <= CONTRACT_TOL
    assert psnr(y, g["y"]) >= {"fp32": 100.0, "fp16": 70.0, "bf16": 46.0}[precision]


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("tag", ["1x32x32", "1x40x56"])
def test_decoder_taps_golden(E, net, golden_dir, tag, precision):
    """decoder stage outputs (models/cdan.py:133,141,149,154) against the reference's own (tests/golden/e2e_dec_taps.npz):
    localises a decoder regression that the final output alone would only detect"""
    z = np.load(os.path.join(golden_dir, "e2e_dec_taps.npz"))
    net.precision = precision
    with torch.no_grad():
        _, taps = net.forward_with_taps(torch.from_numpy(z[f"{tag}:x"]).cuda())
    # (deeper than the encoder taps and, at 32x32 input, only 4x4 pixels wide at the bottleneck: measured bf16 1.3e-2, fp16 1.5e-3)
    tol = {"fp32": FP32_TOL, "bf16": 2e-2, "fp16": 2.5e-3}[precision]
    for k in ("dec1", "dec2", "dec3", "dec4"):
        ref = torch.from_numpy(z[f"{tag}:{k}"])
        got = taps[k][:, :ref.shape[1]]
        err = rel_to_max(got, ref)
        assert err <= tol, f"{k}: {err:.3e}"


@pytest.fixture(scope="module")
def oracle_256():
    """(input, oracle output) at BASELINE configs[1]'s image size, computed once for both precisions"""
    from oracle import cdan_oracle as O
    from oracle import params as P
    x, _ = P.lowlight_batch(5, 1, 256, 256)       # one image: the CPU oracle dominates this suite's run time
    with torch.no_grad():
        return x, O.cdan_forward(P.make_state_dict(42), x)


@pytest.mark.parametrize("precision", PRECISIONS)
def test_e2e_oracle_256(E, net, oracle_256, precision):
    """BASELINE configs[1] image size (256x256, low-light recipe) against the oracle."""
    x, ref = oracle_256
    with torch.no_grad():
        net.precision = precision
        y = net(x.cuda())
    err = rel_to_max(y, ref)
    print(f"[{precision}] 1x3x256x256: rel-to-max {err:.3e}  PSNR vs oracle {psnr(y, ref):.1f} dB")
    assert err <= out_tol_for(precision)


def test_non_multiple_of_8_rejected(E, L, net):
    with pytest.raises(L.MdieError):
        net(torch.rand(1, 3, 36, 36, device="cuda"))


def test_cpu_input_rejected(E, L, net):
    with pytest.raises(L.MdieError):
        net(torch.rand(1, 3, 32, 32))


@pytest.mark.parametrize("precision", PRECISIONS)
def test_full_batch_properties(E, net, precision):
    """BASELINE configs[1] at full size (B=32, 256x256): size-independent properties --
    images are independent in eval mode (running-stat BN, per-image CBAM pools), so any
    image of the batch must equal the same image run alone, bit for bit, and a repeated
    run must be bitwise identical (no atomics on the path)."""
    from oracle import params as P
    B = 32                                  # BASELINE configs[1] itself, in every storage type
    x, _ = P.lowlight_batch(3, B, 256, 256)
    x = x.cuda()
    net.precision = precision
    with torch.no_grad():
        y = net(x)
        y2 = net(x)
        assert torch.equal(y, y2)
        for i in (0, B // 2 + 1, B - 1):
            yi = net(x[i:i + 1])
            assert torch.equal(yi[0], y[i]), f"image {i} depends on its batch"
    assert torch.isfinite(y).all() and y.min() > 0 and y.max() < 1


@pytest.mark.parametrize("precision", PRECISIONS)
def test_every_image_is_independent_of_its_batch_at_every_stage(E, precision):
    """Bitwise batch independence, all of it: the batch of 32 at 256x256 (every storage type) cut into sub-batches of 8, 4 and 1
    at EVERY position -- each sub-batch through an engine of its own -- must reproduce its images of the whole batch bit for bit
    in the output and in every stage tap.  (Round 3 found images 16 and 28 of this very batch one bf16 ulp apart, in ONE
    element of the bottleneck CBAM's output, between a batch of 32 and a batch of 8: the tile edge of the pooling partials
    the convolution in front of that CBAM emits followed the batch size, so the pooled average was summed in another
    grouping -- mdie_conv_tile is a function of the map alone since.  Checking three images of one batch had missed it.)"""
    from mdie_amd import engine as EG
    from oracle import params as P
    dev = torch.device("cuda", 0)
    sd = P.make_state_dict(42)
    B = 32
    x = P.lowlight_batch(1, B, 256, 256)[0].to(dev)
    with torch.no_grad():
        whole = EG.CdanEngine(dev, precision).load(sd)
        yw, exw = whole.forward(x, want_taps=True)
        for nb in (8, 4, 1):
            eng = EG.CdanEngine(dev, precision).load(sd)
            for i0 in (range(0, B, nb) if nb > 1 else [i for i in (0, 5, 16, 28, B - 1) if i < B]):
                y, ex = eng.forward(x[i0:i0 + nb].contiguous(), want_taps=True)
                bad = [n for n in ex["taps"] if not torch.equal(ex["taps"][n], exw["taps"][n][i0:i0 + nb])]
                assert not bad and torch.equal(y, yw[i0:i0 + nb]), f"images {i0}..{i0 + nb - 1} alone differ from the batch at {bad or 'the output'}"


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
@pytest.mark.parametrize("shape", [(32, 256, 256), (1, 256, 256), (3, 48, 80)])
def test_folded_tail_equals_general_chain_in_the_network(E, net, precision, shape):
    """The whole forward with decoder.final_dense's transition folded into its producers (what 16-bit engines run on
    pictures of whole 16x16 tiles) against the same forward with MDIE_FWD_GENERAL_TAIL (the chain fp32 and ragged extents
    take): identical up to the order of one fp32 sum over five segments -- and one image alone runs the same fold as the
    batch of 32 (the choice never depends on B), so bitwise batch independence is kept by construction."""
    from oracle import params as P
    x = P.lowlight_batch(31, *shape)[0].cuda()
    net.precision = precision
    eng = net._engine(x.device)
    with torch.no_grad():
        y_fold = eng.forward(x, out=torch.empty_like(x))
        y_gen = eng.forward(x, out=torch.empty_like(x), general_tail=True)
        if shape[0] == 32:
            one = eng.forward(x[5:6].contiguous(), out=torch.empty_like(x[5:6]))
            assert torch.equal(one[0], y_fold[5])
    assert (y_fold - y_gen).abs().max().item() <= 3e-6


def test_checkpoint_update_is_picked_up(E, net):
    from oracle import params as P
    x = torch.rand(1, 3, 32, 32, device="cuda")
    net.precision = "fp32"
    with torch.no_grad():
        y0 = net(x)
        net.load_state_dict(P.make_state_dict(43), strict=True)
        y1 = net(x)
        net.load_state_dict(P.make_state_dict(42), strict=True)
        y2 = net(x)
    assert not torch.equal(y0, y1)
    assert torch.equal(y0, y2)


# ---------------------------------------------------------------------------------------------------------------------
# per-op vectors (ConvBlock, DenseBlock, CBAM, ConvTranspose2d, bilinear x2 + add)
# ---------------------------------------------------------------------------------------------------------------------
def _dt(E, precision):
    return E.dtype_id(precision)


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("cin,cout", [(16, 32), (3, 16), (32, 64)])
def test_conv_block(E, L, golden_dir, cin, cout, precision):
    g, p = _golden(golden_dir, f"op_convblock_{cin}_{cout}.npz")
    dt = _dt(E, precision)
    cin_st = (cin + 15) // 16 * 16
    s, t = bn_fold(p, "bn")
    w = E.pack_conv_weight(p["conv.weight"], dt, cin_stored=cin_st).cuda()
    x = E.to_nhwc(pad_c(g["x"], cin_st).cuda(), dt)
    for pool, key in ((False, "y"), (True, "y_pool")):
        y = E.conv_fwd([x], w, s.cuda(), (p["conv.bias"] * s + t).cuda(), dtype=dt, ksize=3, cout=cout,
                       act=L.ACT_RELU, pool=pool)
        err = rel_to_max(E.to_nchw(y, dt), g[key])
        assert err <= tol_for(precision), f"pool={pool}: {err:.3e}"


@pytest.mark.parametrize("precision", PRECISIONS)
def test_first_layer_kernel(E, L, golden_dir, precision):
    """ConvBlock(3,16) through the im2col first-layer kernel (fp32 NCHW in, NHWC out)."""
    g, p = _golden(golden_dir, "op_convblock_3_16.npz")
    dt = _dt(E, precision)
    s, t = bn_fold(p, "bn")
    for pool, key in ((False, "y"), (True, "y_pool")):
        y = E.conv_first_fwd(g["x"].cuda(), p["conv.weight"], s.cuda(), (p["conv.bias"] * s + t).cuda(), dtype=dt,
                             act=L.ACT_RELU, pool=pool)
        err = rel_to_max(E.to_nchw(y, dt), g[key])
        assert err <= tol_for(precision), f"pool={pool}: {err:.3e}"


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
@pytest.mark.parametrize("shape,cout", [((3, 40, 24), 64), ((2, 64, 48), 64), ((9, 32, 32), 64), ((2, 48, 80), 128), ((3, 40, 24), 128), ((40, 64, 64), 64)])
def test_first_layer_pooled_kernel(E, L, precision, shape, cout):
    """conv_first_pool_kernel (csrc/conv.hip: encoder.conv1 as the network runs it -- 16-bit, 64 outputs per workgroup, ReLU +
    2x2 max-pool, MFMA operand roles exchanged, persistent over tile runs) against torch's CPU convolution + max_pool2d, and
    bit for bit against the generic first-layer kernel: the same call without pooling, pooled afterwards (max of rounded
    values = rounding of the max).  Ragged tiles (40x24), several 64-channel tiles per pixel tile (cout 128), tile runs that
    cross image boundaries, and 640 tiles on the persistent grid."""
    import torch.nn.functional as F
    dt, td = E.dtype_id(precision), TORCH_DT[precision]
    B, H, W = shape
    g = torch.Generator().manual_seed(H * 7 + cout)
    x = torch.rand(B, 3, H, W, generator=g)
    w = torch.randn(cout, 3, 3, 3, generator=g) * 0.3
    sc, sh = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.2
    y = E.conv_first_fwd(x.cuda(), w, sc.cuda(), sh.cuda(), dtype=dt, act=L.ACT_RELU, pool=True)
    full = E.conv_first_fwd(x.cuda(), w, sc.cuda(), sh.cuda(), dtype=dt, act=L.ACT_RELU, pool=False)
    torch.cuda.synchronize()
    pooled = F.max_pool2d(full.float().permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1).to(td)
    assert torch.equal(y, pooled), "pooled first-layer kernel vs generic kernel + max_pool2d"
    rnd = lambda t: t.to(td).float()
    ref = F.max_pool2d(torch.relu(F.conv2d(rnd(x), rnd(w), padding=1) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)), 2, 2).permute(0, 2, 3, 1)
    assert rel_to_max(y, ref) <= {"bf16": 8e-3, "fp16": 1e-3}[precision]


def _dense_block(E, L, p, x_nchw, cin, dt):
    c0 = (cin + 15) // 16 * 16
    gap = c0 - cin
    dev = "cuda"
    base = E.to_nhwc(pad_c(x_nchw, c0).to(dev), dt)
    segs = [base]
    for i in range(4):
        c = cin + 16 * i
        s, t = bn_fold(p, f"layers.{i}.0")
        pre_s, pre_t = torch.zeros(c0 + 16 * i), torch.zeros(c0 + 16 * i)
        idx = torch.tensor([k + (gap if k >= cin else 0) for k in range(c)])
        pre_s[idx], pre_t[idx] = s, t
        w = E.pack_conv_weight(p[f"layers.{i}.2.weight"], dt, cin_stored=c0 + 16 * i, split=cin, gap=gap).to(dev)
        segs.append(E.conv_fwd(segs, w, torch.ones(16, device=dev), p[f"layers.{i}.2.bias"].to(dev), dtype=dt, ksize=3,
                               cout=16, pre_scale=pre_s.to(dev), pre_shift=pre_t.to(dev)))
    c = cin + 64
    s, t = bn_fold(p, "transition_layer.0")
    pre_s, pre_t = torch.zeros(c0 + 64), torch.zeros(c0 + 64)
    idx = torch.tensor([k + (gap if k >= cin else 0) for k in range(c)])
    pre_s[idx], pre_t[idx] = s, t
    w = E.pack_conv_weight(p["transition_layer.2.weight"], dt, cin_stored=c0 + 64, cout_stored=c0, split=cin, gap=gap).to(dev)
    y = E.conv_fwd(segs, w, pad_v(torch.ones(cin), c0).to(dev), pad_v(p["transition_layer.2.bias"], c0).to(dev), dtype=dt,
                   ksize=1, cout=c0, pre_scale=pre_s.to(dev), pre_shift=pre_t.to(dev))
    return E.to_nchw(y, dt)[:, :cin]


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("cin", [16, 32, 3])
def test_dense_block(E, L, golden_dir, cin, precision):
    g, p = _golden(golden_dir, f"op_denseblock_{cin}.npz")
    y = _dense_block(E, L, p, g["x"], cin, _dt(E, precision))
    err = rel_to_max(y, g["y"])
    assert err <= tol_for(precision), f"{err:.3e}"


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("c", [32, 64, 256])
def test_cbam(E, L, golden_dir, c, precision):
    from models.cbam import CBAM
    g, p = _golden(golden_dir, f"op_cbam_{c}.npz")
    m = CBAM(c, precision=precision)
    m.load_state_dict(p, strict=True)
    m = m.eval().cuda()
    with torch.no_grad():
        y = m(g["x"].cuda())
    err = rel_to_max(y, g["y"])
    assert err <= tol_for(precision), f"cbam: {err:.3e}"
    mc = CBAM(c, no_spatial=True, precision=precision)
    mc.load_state_dict({k: v for k, v in p.items() if k.startswith("ChannelGate")}, strict=True)
    with torch.no_grad():
        yc = mc.eval().cuda()(g["x"].cuda())
    err = rel_to_max(yc, g["y_channel"])
    assert err <= tol_for(precision), f"channel gate: {err:.3e}"


@pytest.mark.parametrize("c,hw", [(64, (24, 40)), (256, (8, 8))])
def test_cbam_in_place_equals_out_of_place(E, L, c, hw):
    """include/mdie.h: `out` of mdie_cbam_fwd may be `x` itself.  Pass 4 batches and prefetches its loads ahead of its stores,
    so this pins that every 16-byte unit is read before the same thread overwrites it: in-place == out-of-place, bit for bit
    (folded gate at C=64, separate gate launch at C=256; with and without the multiplicand)."""
    import ctypes as C
    g = torch.Generator().manual_seed(c)
    for dt, td in ((L.BF16, torch.bfloat16), (L.F32, torch.float32), (L.F16, torch.float16)):
        x = torch.randn(2, *hw, c, generator=g).cuda().to(td)
        mul = torch.randn(2, *hw, c, generator=g).cuda().to(td)
        w1, b1 = torch.randn(c // 16, c, generator=g).cuda() * 0.1, torch.randn(c // 16, generator=g).cuda() * 0.1
        w2, b2 = torch.randn(c, c // 16, generator=g).cuda() * 0.1, torch.randn(c, generator=g).cuda() * 0.1
        w7, bn = torch.randn(2, 7, 7, generator=g).cuda() * 0.1, torch.tensor([0.9, 0.05]).cuda()
        for m in (None, mul):
            ref = E.cbam_fwd(x, w1, b1, w2, b2, w7, bn, dtype=dt, mul=m)
            xi = x.clone()
            n = L.lib.mdie_cbam_workspace_bytes(2, hw[0], hw[1], c)
            ws = torch.empty(n, dtype=torch.uint8, device="cuda")
            d = L.CbamDesc()
            d.dtype, d.B, d.H, d.W, d.C = dt, 2, hw[0], hw[1], c
            d.x, d.x_stride = xi.data_ptr(), c
            d.w1, d.b1, d.w2, d.b2, d.w7, d.bn = (t.data_ptr() for t in (w1, b1, w2, b2, w7, bn))
            d.mul, d.mul_stride = (m.data_ptr(), c) if m is not None else (None, 0)
            d.out, d.out_stride = xi.data_ptr(), c
            d.workspace, d.workspace_bytes = ws.data_ptr(), n
            d.pool_partial, d.pool_slabs = None, 0
            L.check(L.lib.mdie_cbam_fwd(C.byref(d), None), "mdie_cbam_fwd")
            torch.cuda.synchronize()
            assert torch.equal(xi, ref)


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("cin,cout", [(32, 16), (64, 3)])
def test_conv_transpose(E, L, golden_dir, cin, cout, precision):
    g, p = _golden(golden_dir, f"op_convtranspose_{cin}_{cout}.npz")
    dt = _dt(E, precision)
    cst = (cout + 15) // 16 * 16
    w = E.pack_conv_weight(p["weight"], dt, transposed=True, cout_stored=cst).cuda()
    x = E.to_nhwc(g["x"].cuda(), dt)
    y = E.conv_fwd([x], w, pad_v(torch.ones(cout), cst).cuda(), pad_v(p["bias"], cst).cuda(), dtype=dt, ksize=3, cout=cst)
    err = rel_to_max(E.to_nchw(y, dt)[:, :cout], g["y"])
    assert err <= tol_for(precision), f"{err:.3e}"


@pytest.mark.parametrize("precision", PRECISIONS)
def test_up2_add(E, golden_dir, precision):
    g, _ = _golden(golden_dir, "op_up2_add.npz")
    dt = _dt(E, precision)
    y = E.upsample2x_add(E.to_nhwc(g["lo"].cuda(), dt), E.to_nhwc(g["skip"].cuda(), dt), dtype=dt)
    err = rel_to_max(E.to_nchw(y, dt), g["y"])
    assert err <= {"fp32": 1e-6, "bf16": BF16_TOL, "fp16": F16_TOL}[precision], f"{err:.3e}"


def test_up2_add_with_fused_pool(E, L):
    """upsample + skip that also emits the CBAM pooling partials of what it wrote."""
    import ctypes as C
    g = torch.Generator().manual_seed(3)
    for dt, td in ((L.F32, torch.float32), (L.BF16, torch.bfloat16), (L.F16, torch.float16)):
        lo = torch.randn(2, 6, 10, 64, generator=g).cuda().to(td)
        skip = torch.randn(2, 12, 20, 64, generator=g).cuda().to(td)
        ref = E.upsample2x_add(lo, skip, dtype=dt)
        out = torch.empty_like(skip)
        slabs = L.lib.mdie_pool_slabs(12, 20)
        assert slabs == 32 and L.lib.mdie_pool_slabs(512, 512) == 128
        part = torch.zeros(2, slabs, 2, 64, device="cuda")
        L.check(L.lib.mdie_upsample2x_add_pool(dt, 2, 6, 10, 64, lo.data_ptr(), 64, skip.data_ptr(), 64, out.data_ptr(), 64,
                                               part.data_ptr(), slabs, None), "mdie_upsample2x_add_pool")
        # same formula, separately compiled: FMA contraction may differ by an ulp
        assert torch.allclose(out.float(), ref.float(), rtol={L.BF16: 1e-2, L.F16: 1e-3, L.F32: 1e-6}[dt], atol=1e-6)
        o = out.float().reshape(2, -1, 64)
        assert torch.allclose(part[:, :, 0].sum(1), o.sum(1), rtol=1e-5, atol=1e-3)
        assert torch.equal(part[:, :, 1].amax(1), o.amax(1))


def test_upsample_nchw3_last_stage(E, L):
    """Last decoder stage (models/cdan.py:153-154): bilinear x2 of channels 0..2 of an NHWC tensor + the fp32 NCHW network
    input, written as one 16-byte channel group per pixel; taps are read 4 channels at a time, so the source must be
    16-byte aligned with a pixel stride that is a multiple of 4 channels -- anything else is rejected, not mis-read."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(11)
    for dt, td, oc, tol in ((L.F32, torch.float32, 4, 1e-6), (L.BF16, torch.bfloat16, 8, 8e-3), (L.F16, torch.float16, 8, 1e-3)):
        lo = torch.randn(2, 9, 7, 16, generator=g).cuda().to(td)
        x = torch.rand(2, 3, 18, 14, generator=g).cuda()
        out = torch.full((2, 18, 14, oc), 5.0, device="cuda", dtype=td)
        L.check(L.lib.mdie_upsample2x_add_nchw3(dt, 2, 9, 7, lo.data_ptr(), 16, x.data_ptr(), out.data_ptr(), oc, None), "nchw3")
        ref = F.interpolate(lo[..., :3].float().permute(0, 3, 1, 2), scale_factor=2, mode="bilinear", align_corners=False) + x
        assert rel_to_max(out[..., :3].float().permute(0, 3, 1, 2), ref) <= tol
        assert (out[..., 3:] == 0).all()
        assert L.lib.mdie_upsample2x_add_nchw3(dt, 2, 9, 7, lo.data_ptr(), 6, x.data_ptr(), out.data_ptr(), oc, None) != 0
        assert L.lib.mdie_upsample2x_add_nchw3(dt, 2, 9, 7, lo.data_ptr() + 8, 16, x.data_ptr(), out.data_ptr(), oc, None) != 0


@pytest.mark.parametrize("precision", PRECISIONS)
def test_upsample_into_first_dense_layer_fused(E, L, precision):
    """mdie_up_add_dense0_fwd (csrc/updense0.hip): bilinear x2 + x and layer 0 of decoder.final_dense in one launch, K = 27
    im2col'ed into one MFMA step.  The base tensor it writes must equal upsample2x_add_nchw3's bit for bit; g0 is checked
    against torch's CPU convolution of the activated base (ragged extent, zero padding of the ACTIVATED tensor)."""
    import ctypes as C
    import torch.nn.functional as F
    dt, td = E.dtype_id(precision), TORCH_DT[precision]
    vec = 4 if precision == "fp32" else 8
    g = torch.Generator().manual_seed(21)
    B, H, W = 3, 40, 56
    lo = torch.randn(B, H // 2, W // 2, 16, generator=g).cuda().to(td)
    x = torch.rand(B, 3, H, W, generator=g).cuda()
    w = torch.randn(16, 3, 3, 3, generator=g) * 0.3
    ps, pb, bias = torch.rand(8, generator=g) + 0.5, torch.randn(8, generator=g) * 0.3, torch.randn(16, generator=g)
    wp = torch.zeros(L.lib.mdie_conv_first_weight_bytes(dt, 16), dtype=torch.uint8)
    wn = np.ascontiguousarray(w.numpy())
    L.check(L.lib.mdie_pack_conv_first_weight(dt, wn.ctypes.data, 16, 16, wp.data_ptr()), "pack")
    wp, dps, dpb, dbias = wp.cuda(), ps.cuda(), pb.cuda(), bias.cuda()
    base_ref = torch.full((B, H, W, vec), 5.0, device="cuda", dtype=td)
    L.check(L.lib.mdie_upsample2x_add_nchw3(dt, B, H // 2, W // 2, lo.data_ptr(), 16, x.data_ptr(), base_ref.data_ptr(), vec, None), "nchw3")
    base = torch.full((B, H, W, vec), 5.0, device="cuda", dtype=td)
    g0 = torch.full((B, H, W, 32), -7.0, device="cuda", dtype=td)
    d = L.UpDense0Desc()
    d.dtype, d.B, d.H, d.W = dt, B, H, W
    d.lo, d.lo_stride, d.x = lo.data_ptr(), 16, x.data_ptr()
    d.base, d.base_channels, d.weight = base.data_ptr(), vec, wp.data_ptr()
    d.pre_scale, d.pre_shift, d.bias = dps.data_ptr(), dpb.data_ptr(), dbias.data_ptr()
    d.g0, d.g0_stride = g0[..., 8:].data_ptr(), 32        # a 16-channel slice of a wider buffer
    L.check(L.lib.mdie_up_add_dense0_fwd(C.byref(d), None), "mdie_up_add_dense0_fwd")
    torch.cuda.synchronize()
    assert torch.equal(base, base_ref)
    rnd = lambda t: t.to(td).float()
    b3 = base_ref[..., :3].float().cpu().permute(0, 3, 1, 2)
    act = rnd(torch.relu(b3 * ps[:3].view(1, 3, 1, 1) + pb[:3].view(1, 3, 1, 1)))
    ref = F.conv2d(act, rnd(w), bias, padding=1).permute(0, 2, 3, 1)
    assert rel_to_max(g0[..., 8:24], ref) <= {"fp32": 2e-5, "bf16": 8e-3, "fp16": 1e-3}[precision]
    assert (g0[..., :8] == -7.0).all() and (g0[..., 24:] == -7.0).all()


def _final_dense_problem(E, L, precision, shape, signed_scales=False):
    """decoder.final_dense (models/cdan.py:22-53,119,153-157) on seeded parameters, behind three forms of the C ABI:
    run(fold=False): mdie_up_add_dense0_fwd, three 3x3 layers, the 1x1 launch; run(fold=True[, half=True]): the transition folded into the
    four producers (mdie_tr_fuse); run_block(): ONE launch (mdie_final_dense_fwd).  Returns (run, run_block, ref) with ref() = torch's CPU
    arithmetic of the block on a given base tensor."""

    import ctypes as C
    import torch.nn.functional as F
    dt, td = E.dtype_id(precision), TORCH_DT[precision]
    rnd = lambda t: t.to(td).float()
    B, H, W = shape
    g = torch.Generator().manual_seed(H + 3 * B)
    lo = (torch.randn(B, H // 2, W // 2, 16, generator=g) * 0.5).cuda().to(td)
    x = torch.rand(B, 3, H, W, generator=g).cuda()
    # block parameters: 4 layers (pre-activation BN folded, 3x3 conv, bias) + transition (BN over 67 channels, 1x1 conv 67 -> 3, bias)
    ws = [torch.randn(16, 3 + 16 * l, 3, 3, generator=g) * (0.3 if l == 0 else 0.12) for l in range(4)]
    bs = [torch.randn(16, generator=g) * 0.2 for _ in range(4)]
    pss = [torch.rand(3 + 16 * l, generator=g) + 0.5 for l in range(4)]
    pbs = [torch.randn(3 + 16 * l, generator=g) * 0.3 for l in range(4)]
    wt, bt = torch.randn(3, 67, 1, 1, generator=g) * 0.2, torch.randn(3, generator=g) * 0.2
    pst, pbt = torch.rand(67, generator=g) + 0.5, torch.randn(67, generator=g) * 0.3
    if signed_scales:      # a trained BatchNorm's gamma may be negative or (pruned) zero: every third scale flipped, every seventh zero
        for v in pss + [pst]:
            v[1::3] *= -1.0
            v[2::7] = 0.0

    def stored(v, n):          # real channel c >= 3 sits at stored channel c + 5 (the base is one 8-channel group)
        out = torch.zeros(n)
        out[:3] = v[:3]
        out[8:8 + v.numel() - 3] = v[3:]
        return out
    w0 = torch.zeros(L.lib.mdie_conv_first_weight_bytes(dt, 16), dtype=torch.uint8)
    L.check(L.lib.mdie_pack_conv_first_weight(dt, np.ascontiguousarray(ws[0].numpy()).ctypes.data, 16, 16, w0.data_ptr()), "pack")
    w0 = w0.cuda()
    wl = [None] + [E.pack_conv_weight(ws[l], dt, cin_stored=8 + 16 * l, split=3, gap=5).cuda() for l in (1, 2, 3)]
    wtp = E.pack_conv_weight(wt, dt, cout_stored=16, cin_stored=72, split=3, gap=5).cuda()
    ps_d = [stored(pss[l], 8 + 16 * l).cuda() for l in range(4)]
    pb_d = [stored(pbs[l], 8 + 16 * l).cuda() for l in range(4)]
    pst_d, pbt_d = stored(pst, 72).cuda(), stored(pbt, 72).cuda()
    ones = torch.ones(16, device="cuda")
    b_d = [b.cuda() for b in bs]
    bt_d = pad_v(bt, 16).cuda()

    def run(fold, half=False):
        # half: the base stored as HALF a 16-byte group (4 channels per pixel, mdie_seg / base_stride: ABI 24) -- the chain's 16-byte column
        # loads then read the NEXT pixel's bytes into channels 4..7, whose weights are zero.  The bytes behind the LAST pixel are the
        # producer's to define (0 * NaN is NaN): the buffer -- spare pixel included -- starts as NaN, mdie_up_add_dense0_fwd must zero
        # the 8 bytes its consumers will read, and nothing non-finite may reach an output
        bs_ = 4 if half else 8
        base_buf = torch.full((B * H * W + 1, bs_), float("nan"), device="cuda", dtype=td)
        base = base_buf[:B * H * W].view(B, H, W, bs_)
        gs = [torch.full((B, H, W, 16), -7.0, device="cuda", dtype=td) for _ in range(4)]
        y = torch.full((B, 3, H, W), -1.0, device="cuda")
        part = torch.full((B, H, W, 4), 1e9, device="cuda")
        trs = []

        def tr(c0, last=False):
            t = L.TrFuse()
            t.weight, t.c0, t.pre_scale, t.pre_shift = wtp.data_ptr(), c0, pst_d.data_ptr(), pbt_d.data_ptr()
            t.partial_in, t.partial_out = part.data_ptr(), (None if last else part.data_ptr())
            if last:
                t.post_scale, t.post_shift, t.act, t.out_nchw3 = ones.data_ptr(), bt_d.data_ptr(), L.ACT_SIGMOID, y.data_ptr()
            trs.append(t)
            return C.pointer(t)
        u = L.UpDense0Desc()
        u.dtype, u.B, u.H, u.W = dt, B, H, W
        u.lo, u.lo_stride, u.x = lo.data_ptr(), 16, x.data_ptr()
        u.base, u.base_channels, u.base_stride, u.weight = base.data_ptr(), 8, (4 if half else 0), w0.data_ptr()
        u.pre_scale, u.pre_shift, u.bias = ps_d[0].data_ptr(), pb_d[0].data_ptr(), b_d[0].data_ptr()
        u.g0, u.g0_stride = gs[0].data_ptr(), 16
        if fold:
            u.tr = tr(8)
        L.check(L.lib.mdie_up_add_dense0_fwd(C.byref(u), None), "mdie_up_add_dense0_fwd")
        for l in (1, 2, 3):
            d = L.ConvDesc()
            d.dtype, d.B, d.H, d.W, d.ksize, d.nseg = dt, B, H, W, 3, l + 1
            d.inp[0] = L.Seg(base.data_ptr(), 8, bs_)
            for i in range(l):
                d.inp[1 + i] = L.Seg(gs[i].data_ptr(), 16, 16)
            d.cin, d.cout = 8 + 16 * l, 16
            d.pre_scale, d.pre_shift = ps_d[l].data_ptr(), pb_d[l].data_ptr()
            d.weight, d.post_scale, d.post_shift = wl[l].data_ptr(), ones.data_ptr(), b_d[l].data_ptr()
            d.act, d.pool = L.ACT_NONE, 0
            d.out, d.out_stride = gs[l].data_ptr(), 16
            if fold:
                d.tr = tr(8 + 16 * l, last=(l == 3))
            L.check(L.lib.mdie_conv_fwd(C.byref(d), None), "mdie_conv_fwd")
        if not fold:
            d = L.ConvDesc()
            d.dtype, d.B, d.H, d.W, d.ksize, d.nseg = dt, B, H, W, 1, 5
            d.inp[0] = L.Seg(base.data_ptr(), 8, 8)
            for i in range(4):
                d.inp[1 + i] = L.Seg(gs[i].data_ptr(), 16, 16)
            d.cin, d.cout = 72, 16
            d.pre_scale, d.pre_shift = pst_d.data_ptr(), pbt_d.data_ptr()
            d.weight, d.post_scale, d.post_shift = wtp.data_ptr(), ones.data_ptr(), bt_d.data_ptr()
            d.act, d.pool = L.ACT_SIGMOID, 0
            scratch = torch.empty(B, H, W, 16, device="cuda", dtype=td)
            d.out, d.out_stride, d.out_nchw3 = scratch.data_ptr(), 16, y.data_ptr()
            L.check(L.lib.mdie_conv_fwd(C.byref(d), None), "mdie_conv_fwd")
        torch.cuda.synchronize()
        return base, gs, y


    def run_block():
        f = L.FinalDenseDesc()
        f.dtype, f.B, f.H, f.W = dt, B, H, W
        f.lo, f.lo_stride, f.x, f.w0 = lo.data_ptr(), 16, x.data_ptr(), w0.data_ptr()
        for l in (1, 2, 3):
            f.w[l - 1] = wl[l].data_ptr()
        for l in range(4):
            f.pre_scale[l], f.pre_shift[l], f.post_scale[l], f.post_shift[l] = ps_d[l].data_ptr(), pb_d[l].data_ptr(), ones.data_ptr(), b_d[l].data_ptr()
        f.wt, f.tr_pre_scale, f.tr_pre_shift = wtp.data_ptr(), pst_d.data_ptr(), pbt_d.data_ptr()
        f.tr_post_scale, f.tr_post_shift = ones.data_ptr(), bt_d.data_ptr()
        y = torch.full((B, 3, H, W), -1.0, device="cuda")
        f.y = y.data_ptr()
        L.check(L.lib.mdie_final_dense_fwd(C.byref(f), None), "mdie_final_dense_fwd")
        torch.cuda.synchronize()
        return y

    def ref(base):
        # torch CPU arithmetic of the block on the engine's base (stored roundings reproduced: every growth map and every activated operand is rounded to the storage type)
        feats = [base[..., :3].float().cpu().permute(0, 3, 1, 2)]
        for l in range(4):
            cat = torch.cat(feats, 1)
            act = rnd(torch.relu(cat * pss[l].view(1, -1, 1, 1) + pbs[l].view(1, -1, 1, 1)))
            feats.append(rnd(F.conv2d(act, rnd(ws[l]), bs[l], padding=1)))
        cat = torch.cat(feats, 1)
        act = rnd(torch.relu(cat * pst.view(1, -1, 1, 1) + pbt.view(1, -1, 1, 1)))
        return torch.sigmoid(F.conv2d(act, rnd(wt), bt))
    return run, run_block, ref


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
@pytest.mark.parametrize("shape", [(2, 32, 48), (1, 16, 16), (5, 240, 256)])
def test_transition_folded_into_its_producers(E, L, precision, shape):
    """decoder.final_dense with its transition (BN -> ReLU -> Conv1x1 67 -> 3 -> sigmoid, models/cdan.py:48-53,155-157) folded
    into the four producers of the transition's input (mdie_tr_fuse: csrc/updense0.hip, csrc/conv_thin.hip) against
      * the SAME block as the general chain -- mdie_up_add_dense0_fwd, three 3x3 layers, the 1x1 launch -- whose terms are the
        same roundings of the same stored tensors, so the two may differ only in the order of the fp32 sum over the five
        segments: <= 3e-6 on outputs in (0, 1), growth maps g0..g2 bit-identical, and
      * torch's CPU arithmetic of the block on the engine's own base tensor.
    A few tiles (one per workgroup), a single tile, and runs of several tiles per persistent workgroup across image borders."""
    run, _, ref_of = _final_dense_problem(E, L, precision, shape)
    base_u, gs_u, y_u = run(False)
    base_f, gs_f, y_f = run(True)
    base_h, gs_h, y_h = run(True, half=True)
    assert torch.equal(base_h[..., :3], base_f[..., :3]) and (base_h[..., 3] == 0).all()
    assert torch.isfinite(y_h).all(), "the NaN behind the half-group buffer's last pixel reached the output: its producer must zero those 8 bytes"
    assert torch.equal(y_h, y_f) and all(torch.equal(gs_h[l], gs_f[l]) for l in range(3)), "the half-group base must not change a bit"
    assert torch.equal(base_u, base_f)
    for l in range(3):
        assert torch.equal(gs_u[l], gs_f[l]), f"growth map {l} must not change"
    assert (gs_f[3] == -7.0).all(), "the last growth map is never stored when the transition is folded in"
    assert (y_f - y_u).abs().max().item() <= 3e-6
    assert rel_to_max(y_f, ref_of(base_f)) <= {"bf16": 8e-3, "fp16": 1e-3}[precision]


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
@pytest.mark.parametrize("shape", [(2, 32, 48), (1, 16, 16), (1, 8, 16), (3, 24, 32), (5, 240, 256), (33, 64, 48)])
def test_final_block_one_launch_equals_the_chain(E, L, precision, shape):
    """mdie_final_dense_fwd (csrc/final_block.hip, ABI 27): upsample + x, the four DenseBlock layers, the transition and the sigmoid in ONE
    launch -- 16 x 8 output tiles, growth maps in LDS, halo rings recomputed -- against the chain of launches on the same parameters.
    The kernel keeps the chain's arithmetic operation for operation, so y must be BIT-IDENTICAL to the folded chain's (H a multiple of 16:
    conv_thin's tiles) and within the order of one fp32 sum (3e-6) of the general chain's everywhere; and <= the storage type's bound of
    torch's CPU arithmetic of the block.  Single tiles, pictures whose every tile touches a border, H a multiple of 8 only, runs of many
    tiles per persistent workgroup across picture and image borders, more tiles than workgroups (33 images)."""
    run, run_block, ref_of = _final_dense_problem(E, L, precision, shape)
    B, H, W = shape
    base_u, gs_u, y_u = run(False)
    y_b = run_block()
    assert torch.isfinite(y_b).all() and (y_b > 0).all() and (y_b < 1).all(), "every output pixel must have been written with a sigmoid value"
    assert (y_b - y_u).abs().max().item() <= 3e-6
    if H % 16 == 0 and W % 16 == 0:
        _, _, y_f = run(True, half=True)
        assert torch.equal(y_b, y_f), f"one launch vs folded chain: {(y_b != y_f).sum().item()} of {y_b.numel()} values differ, max {(y_b - y_f).abs().max().item():.3e}"
    assert torch.equal(run_block(), y_b), "run-to-run"
    assert rel_to_max(y_b, ref_of(base_u)) <= {"bf16": BF16_TOL, "fp16": F16_TOL}[precision]     # (torch CPU arithmetic: the operator bound)


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_final_block_with_negative_and_zero_batchnorm_scales(E, L, precision):
    """The one-launch block against the chain when a third of the folded BatchNorm scales are negative and some are zero (a trained gamma may be
    either): the block keeps the chain's pre-activation arithmetic, so nothing about it depends on a scale's sign -- bit-identical again."""
    run, run_block, ref_of = _final_dense_problem(E, L, precision, (3, 48, 64), signed_scales=True)
    base_u, _, y_u = run(False)
    _, _, y_f = run(True, half=True)
    y_b = run_block()
    assert torch.equal(y_b, y_f) and (y_b - y_u).abs().max().item() <= 3e-6
    assert rel_to_max(y_b, ref_of(base_u)) <= {"bf16": BF16_TOL, "fp16": F16_TOL}[precision]


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
@pytest.mark.parametrize("shape", [(2, 64, 64), (3, 128, 160), (32, 256, 256)])
def test_network_with_the_one_launch_block_equals_the_chain(E, precision, shape):
    """MDIE_FWD_BLOCK_TAIL: mdie_cdan_forward runs decoder.final_dense as ONE launch where the folded chain of four launches would run (16-bit
    types, whole 16x16 tiles, one weight set, no taps).  Same arithmetic: the network's output must not change by a bit."""
    from oracle import params as P
    x, _ = P.lowlight_batch(45, *shape)
    x = x.cuda()
    eng = E.CdanEngine("cuda", precision).load(P.make_state_dict(42))
    y_block = eng.forward(x, out=torch.empty_like(x), block_tail=True).clone()
    y_chain = eng.forward(x, out=torch.empty_like(x)).clone()
    torch.cuda.synchronize()
    assert torch.equal(y_block, y_chain), f"{(y_block != y_chain).sum().item()} values differ, max {(y_block - y_chain).abs().max().item():.3e}"
    _, ex_b = eng.forward(x, out=torch.empty_like(x), profile=True, block_tail=True)
    _, ex_c = eng.forward(x, out=torch.empty_like(x), profile=True)
    assert len(ex_c["launches"]) - len(ex_b["launches"]) == 3, "four launches of the chain become one"
    assert abs(sum(b for _, b, _ in ex_b["launch_info"]) - sum(b for _, b, _ in ex_c["launch_info"])) <= 1e-6 * sum(b for _, b, _ in ex_c["launch_info"]), \
        "the block is booked at the chain's share of the SURVEY 8d byte model"


def test_forward_of_a_new_shape_does_not_synchronise_or_time_anything(E):
    """CdanEngine.forward is asynchronous like the C ABI under it: the FIRST forward of a new batch shape enqueues its launches behind whatever
    the stream holds and returns (round 5 timed 652 forwards inside it -- CdanEngine.tune, now an explicit call bench.py makes in its
    warm-up).  A long busy-wait kernel is put on the stream first: when forward() returns, the stream must still be busy."""
    from oracle import params as P
    eng = E.CdanEngine("cuda", "bf16").load(P.make_state_dict(42))
    shape = (4, 256, 512)      # a shape no other test uses: conv_wide takes encoder.conv4, so round 5's forward would have tuned here
    assert E._share_cu_eligible(eng.dtype, *shape) and eng._key(*shape) not in E._SHARE_CU
    x, _ = P.lowlight_batch(46, *shape)
    x = x.cuda()
    y = torch.empty_like(x)
    eng._workspace(*shape)        # (the arena allocation may synchronise; it is not the forward)
    torch.cuda.synchronize()
    torch.cuda._sleep(int(2.0e8))     # ~0.1 s of GPU time in front of the forward
    eng.forward(x, out=y)
    assert not torch.cuda.current_stream().query(), "forward() returned only after the stream had drained: it synchronised"
    torch.cuda.synchronize()
    assert eng._key(*shape) not in E._SHARE_CU and eng.form(*shape) == E.DEFAULT_FORM
    _, ex = eng.forward(x, out=y, profile=True)
    assert len(ex["launches"]) == 39, len(ex["launches"])


def test_transition_fusion_rejects_what_it_cannot_run(E, L):
    """mdie_tr_fuse is built for 16-bit types on whole 16x16 tiles: everything else is refused loudly (the engine runs the general chain there);
    so is a half-group segment (8 channels at stride 4) outside the folded chain"""
    import ctypes as C
    x = torch.zeros(1, 24, 32, 16, device="cuda")
    d = L.ConvDesc()
    d.dtype, d.B, d.H, d.W, d.ksize, d.nseg = L.BF16, 1, 16, 16, 3, 2
    d.inp[0], d.inp[1] = L.Seg(x.data_ptr(), 8, 4), L.Seg(x.data_ptr(), 16, 16)
    d.cin, d.cout = 24, 16
    wv = torch.zeros(L.lib.mdie_conv_weight_bytes(L.BF16, 3, 16, 24), dtype=torch.uint8, device="cuda")
    vv = torch.zeros(80, device="cuda")
    d.pre_scale, d.pre_shift, d.weight, d.post_scale, d.post_shift = vv.data_ptr(), vv.data_ptr(), wv.data_ptr(), vv.data_ptr(), vv.data_ptr()
    d.out, d.out_stride = x.data_ptr(), 16
    assert L.lib.mdie_conv_fwd(C.byref(d), None) == -1 and "stride" in L.lib.mdie_last_error().decode()      # no tr: stride < channels is invalid
    w = torch.zeros(L.lib.mdie_conv_weight_bytes(L.BF16, 3, 16, 16), dtype=torch.uint8, device="cuda")
    v = torch.zeros(80, device="cuda")
    t = L.TrFuse()
    t.weight, t.c0, t.pre_scale, t.pre_shift, t.partial_in, t.partial_out = w.data_ptr(), 8, v.data_ptr(), v.data_ptr(), v.data_ptr(), v.data_ptr()
    for dt, H in ((L.F32, 32), (L.BF16, 24)):       # fp32; a height that is not a multiple of 16
        d = L.ConvDesc()
        d.dtype, d.B, d.H, d.W, d.ksize, d.nseg = dt, 1, H, 32, 3, 1
        d.inp[0] = L.Seg(x.data_ptr(), 16, 16)
        d.cin, d.cout = 16, 16
        d.pre_scale, d.pre_shift, d.weight, d.post_scale, d.post_shift = v.data_ptr(), v.data_ptr(), w.data_ptr(), v.data_ptr(), v.data_ptr()
        d.out, d.out_stride = x.data_ptr(), 16
        d.tr = C.pointer(t)
        assert L.lib.mdie_conv_fwd(C.byref(d), None) == -1      # MDIE_EINVAL


def test_conv_rejects_bad_arguments(E, L):
    x = torch.zeros(1, 4, 4, 16, device="cuda")
    w = torch.zeros(L.lib.mdie_conv_weight_bytes(L.F32, 3, 16, 16), dtype=torch.uint8, device="cuda")
    v = torch.zeros(16, device="cuda")
    with pytest.raises(L.MdieError):
        E.conv_fwd([x[..., :6]], w, v, v, dtype=L.F32, ksize=3, cout=16)       # segment is not whole 16-byte channel groups
    with pytest.raises(L.MdieError):
        E.conv_fwd([x], w, v, v, dtype=L.F32, ksize=5, cout=16)                # unsupported kernel size
    with pytest.raises(L.MdieError):
        E.conv_fwd([x[:, :3, :3]], w, v, v, dtype=L.F32, ksize=3, cout=16, pool=True)  # odd extent with pool


# ---------------------------------------------------------------------------------------------------------------------
# around the network: feed, post-processing, uint8 output, PSNR / SSIM
# ---------------------------------------------------------------------------------------------------------------------
def test_postprocessing_matches_reference_vectors(E, golden_dir):
    import mdie_amd.pipeline as PL
    z = np.load(os.path.join(golden_dir, "op_postproc.npz"))
    y = torch.from_numpy(z["y"]).cuda()
    one = lambda name, **a: {"enabled": True, "ops": [{"name": name, "args": a}]}
    cases = {"contrast": one("enhance_contrast", contrast_factor=1.03), "color": one("enhance_color", saturation_factor=1.55),
             "sharpen": one("sharpen", strength=0.5), "denoise": one("soft_denoise", sigma=0.2),
             "chain_lowlight": {"enabled": True, "ops": [{"name": "enhance_contrast", "args": {"contrast_factor": 1.03}},
                                                         {"name": "enhance_color", "args": {"saturation_factor": 1.55}}]},
             "chain4": {"enabled": True, "ops": [{"name": "soft_denoise", "args": {"sigma": 0.3}}, {"name": "sharpen", "args": {"strength": 0.7}},
                                                 {"name": "enhance_contrast", "args": {"contrast_factor": 1.2}},
                                                 {"name": "enhance_color", "args": {"saturation_factor": 0.8}}]}}
    for key, cfg in cases.items():
        out = PL.apply_postprocessing(y, cfg)
        err = (out.cpu() - torch.from_numpy(z[key])).abs().max().item()
        assert err <= 2e-6, f"{key}: {err:.3e}"
    out, u8 = PL.apply_postprocessing(y, cases["chain_lowlight"], want_uint8=True)
    ref_u8 = torch.from_numpy(z["chain_lowlight_u8"])
    diff = (u8.cpu().int() - ref_u8.int()).abs()
    # truncation of x*255 can flip by one where the fp32 value sits within an ulp of an integer
    assert diff.max().item() <= 1 and (diff > 0).float().mean().item() < 1e-3
    assert PL.apply_postprocessing(y, {"enabled": False}) is y
    with pytest.raises(ValueError):
        PL.apply_postprocessing(y, {"enabled": True, "ops": [{"name": "posterize"}]})


def test_feed_and_uint8_roundtrip(E):
    import mdie_amd.pipeline as PL
    g = torch.Generator().manual_seed(5)
    img = torch.randint(0, 256, (3, 24, 40, 3), generator=g, dtype=torch.uint8)
    x = PL.feed_uint8(img.cuda())
    assert torch.equal(x.cpu(), img.permute(0, 3, 1, 2).float() / 255.0)
    back = PL.to_uint8_hwc(x)
    # (k/255)*255 truncates to k-1 for the k where the product lands just below k, exactly as numpy does
    expect = torch.from_numpy(((img.float() / 255.0).numpy() * 255.0).clip(0, 255).astype(np.uint8))
    assert torch.equal(back.cpu(), expect)


@pytest.mark.parametrize("shape", [(2, 3, 32, 48), (4, 3, 64, 64), (1, 3, 17, 23), (8, 3, 256, 256)])
def test_psnr_ssim_vs_oracle(E, shape):
    import mdie_amd.pipeline as PL
    from oracle import metrics_oracle as M
    g = torch.Generator().manual_seed(shape[0] * 100 + shape[2])
    t = torch.rand(shape, generator=g)
    p = (t + 0.05 * torch.randn(shape, generator=g)).clamp(0, 1)
    out = PL.psnr_ssim(p.cuda(), t.cuda()).cpu()
    assert out[0].item() == pytest.approx(M.psnr(p, t), abs=2e-3)     # dB
    assert out[1].item() == pytest.approx(M.ssim(p, t), abs=2e-5)


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE configs[0]: a noise-style config, 64x64, batch 4, driven through run.py's main()
# ---------------------------------------------------------------------------------------------------------------------
def test_run_py_test_phase_end_to_end(E, tmp_path):
    import json as _json
    from PIL import Image
    from oracle import cdan_oracle as O
    from oracle import params as P
    from mdie_amd import host as H

    root = str(tmp_path)
    clean, _ = P.lowlight_batch(77, 6, 64, 64)
    _, clean = P.lowlight_batch(77, 6, 64, 64)
    g = torch.Generator().manual_seed(9)
    noisy = (clean + 0.08 * torch.randn(clean.shape, generator=g)).clamp(0, 1)
    for sub, data in (("degraded", noisy), ("clean", clean)):
        os.makedirs(os.path.join(root, "data", sub))
        for i in range(6):
            a = (data[i].permute(1, 2, 0).numpy() * 255).round().astype(np.uint8)
            Image.fromarray(a).save(os.path.join(root, "data", sub, f"{i:03d}.png"))
    sd = P.make_state_dict(42)
    os.makedirs(os.path.join(root, "weights"))
    torch.save(sd, os.path.join(root, "weights", "CDAN_noise.pt"))    # a "reference-trained" checkpoint: same 236 keys

    cfg = H.load_config(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "config", "example_noise_64.json"), "test")
    cfg["test"]["dataset"]["args"]["input_root"] = os.path.join(root, "data", "degraded")
    cfg["test"]["dataset"]["args"]["target_root"] = os.path.join(root, "data", "clean")
    cfg["test"]["model_path"] = os.path.join(root, "weights")
    cfg["save_outputs"]["output_dir"] = os.path.join(root, "out")
    cfg["logging"]["root_dir"] = os.path.join(root, "runs")
    with pytest.warns(UserWarning, match="lpips"):
        model = H.run(cfg)

    res = model.results
    assert res["n_images"] == 6 and {"psnr", "ssim"} <= set(res["raw"]) and {"psnr", "ssim"} <= set(res["post"])
    assert "loss_total" in res["raw"]          # the loss pipeline is evaluated in the test phase too (models/model.py:257)
    outs = sorted(os.listdir(os.path.join(root, "out")))
    assert len(outs) == 12 and outs[0] == "pp_1.png" and outs[-1] == "raw_6.png"     # the reference's names (models/model.py:90)
    # the saved raw image of sample 0 == the oracle on the same decoded uint8 input
    x0 = torch.from_numpy(np.asarray(Image.open(os.path.join(root, "data", "degraded", "000.png")))).permute(2, 0, 1).float()[None] / 255.0
    with torch.no_grad():
        ref = O.cdan_forward(sd, x0)
    ref_u8 = (ref[0].permute(1, 2, 0).numpy() * 255).clip(0, 255).astype(np.uint8)
    got = np.asarray(Image.open(os.path.join(root, "out", "raw_1.png")))
    assert np.abs(got.astype(int) - ref_u8.astype(int)).max() <= 1
    run_dirs = os.listdir(os.path.join(root, "runs", "noise_example"))
    assert len(run_dirs) == 1 and os.path.exists(os.path.join(root, "runs", "noise_example", run_dirs[0], "summary.json"))


@pytest.mark.parametrize("prec,cin_segs", [("bf16", [8]), ("bf16", [8, 16]), ("bf16", [16, 8, 8]), ("fp32", [8]), ("fp32", [4, 8, 4]),
                                           ("fp16", [8]), ("fp16", [16, 8, 8])])
@pytest.mark.parametrize("shape,pre", [((4, 200, 216), True), ((5, 176, 160), False)])
def test_thin_single_chunk_conv(E, L, prec, cin_segs, shape, pre):
    """3x3 convolution with 16 outputs whose input fits one K chunk (the first DenseLayers of final_dense, models/cdan.py:150)
    on a grid large enough for 16x16 tiles (>= 512 workgroups).  Ragged tile edges, several input segments with
    their own strides, pre-activation BN+ReLU, output written into a slice of a wider buffer; checked against torch's CPU
    convolution on the same (bf16-rounded where stored as bf16) operands."""
    import ctypes as C
    import torch.nn.functional as F
    bf = prec != "fp32"       # 16-bit storage: operands rounded to the stored type first
    dt, td = E.dtype_id(prec), TORCH_DT[prec]
    rnd = (lambda t: t.to(td).float())
    B, H, W = shape
    cin, cout = sum(cin_segs), 16
    g = torch.Generator().manual_seed(cin * 7 + H)
    strides = [c + (8 if i % 2 else 0) for i, c in enumerate(cin_segs)]
    bufs = [rnd(torch.randn(B, H, W, st, generator=g)) for st in strides]
    w = rnd(torch.randn(cout, cin, 3, 3, generator=g) * 0.2)
    sc, sh = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    ps, pt = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.3
    x = torch.cat([b[..., :c] for b, c in zip(bufs, cin_segs)], 3)
    xa = rnd(torch.relu(x * ps + pt)) if pre else x
    ref = torch.relu(F.conv2d(xa.permute(0, 3, 1, 2), w, padding=1) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)).permute(0, 2, 3, 1)

    dbufs = [b.cuda().to(td) for b in bufs]
    wp = E.pack_conv_weight(w, dt, cin_stored=cin).cuda()
    dsc, dsh, dps, dpt = sc.cuda(), sh.cuda(), ps.cuda(), pt.cuda()
    out = torch.full((B, H, W, 40), -7.0, device="cuda", dtype=td)
    d = L.ConvDesc()
    d.dtype, d.B, d.H, d.W, d.ksize, d.nseg = dt, B, H, W, 3, len(dbufs)
    for i, (b, c) in enumerate(zip(dbufs, cin_segs)):
        d.inp[i] = L.Seg(b.data_ptr(), c, b.shape[3])
    d.cin, d.cout = cin, cout
    d.pre_scale, d.pre_shift = (dps.data_ptr(), dpt.data_ptr()) if pre else (None, None)
    d.weight, d.post_scale, d.post_shift = wp.data_ptr(), dsc.data_ptr(), dsh.data_ptr()
    d.act, d.pool = L.ACT_RELU, 0
    d.out, d.out_stride = out[..., 16:].data_ptr(), 40
    L.check(L.lib.mdie_conv_fwd(C.byref(d), None), "mdie_conv_fwd")
    torch.cuda.synchronize()
    assert rel_to_max(out[..., 16:32], ref) <= {"fp32": 2e-5, "bf16": 8e-3, "fp16": 1e-3}[prec]
    assert (out[..., :16] == -7.0).all() and (out[..., 32:] == -7.0).all()     # nothing written outside the slice


@pytest.mark.parametrize("prec", ["bf16", "fp16"])
@pytest.mark.parametrize("shape", [(32, 256, 256), (8, 256, 256), (2, 256, 512), (3, 64, 64)])
def test_conv4_forms_are_bit_identical(E, L, prec, shape):
    """mdie_conv_desc.share_cu / MDIE_FWD_SHARE_CU_CONV4 / MDIE_FWD_YIELD_CU_CONV4: encoder.conv4 + BN + ReLU with the pooling partials of
    the bottleneck CBAM (models/cdan.py:95-96, models/cbam.py:41,44) as conv_wide with one run of items per CU, with two shorter runs, or
    on conv_kernel.  The host picks between the three by TIMING them (CdanEngine.tune), so they must agree bit for bit in the output
    and in every stage tap -- also where conv_wide does not apply and the flags change nothing (3 x 64 x 64).  And tune() must come
    back with a decision it remembers."""
    from oracle import params as P
    sd = P.make_state_dict(42)
    x, _ = P.lowlight_batch(44, *shape)
    x = x.cuda()
    eng = E.CdanEngine("cuda", prec).load(sd)
    outs = {}
    for form in (0, 1, 2, 3):
        eng.share_cu = form
        y, ex = eng.forward(x, want_taps=True)
        outs[form] = (y.clone(), {k: v.clone() for k, v in ex["taps"].items()})
    torch.cuda.synchronize()
    for form in (1, 2, 3):
        assert torch.equal(outs[0][0], outs[form][0]), form
        for k in outs[0][1]:
            assert torch.equal(outs[0][1][k], outs[form][1][k]), (form, k)
    eng.share_cu = None
    # forward() runs the static default untimed; tune() is the explicit call that times the forms and is remembered for the shape
    assert eng.form(*shape) == (E.DEFAULT_FORM if E._share_cu_eligible(eng.dtype, *shape) else 0) or eng._key(*shape) in E._SHARE_CU
    d1 = eng.tune(x)
    d2 = eng.tune(x)
    assert d1 == d2 and d1 in (0, 1, 2, 3) and eng.form(*shape) == d1
    assert torch.equal(eng.forward(x), outs[0][0])
    if not E._share_cu_eligible(eng.dtype, *shape):
        assert d1 == 0


@pytest.mark.parametrize("prec", ["bf16", "fp16", "fp32"])
@pytest.mark.parametrize("shape", [(2, 64, 64), (1, 40, 72), (3, 128, 160)])
def test_forward_does_not_depend_on_what_the_workspace_held(E, prec, shape):
    """The workspace arena is `torch.empty` and outlives plans: every byte a launch READS must have been written by an earlier launch of the
    same forward.  The arena is filled with 0xFF (NaN in all three element types; as fp32 partials too) and then with 0x7F / 0x00 between
    forwards: the outputs must be bit-identical and finite.  Covers the half-group base of decoder.final_dense (whose last 16-byte load
    reads 8 bytes behind the last stored pixel: zero weights do not neutralise a NaN), pooled partials, the transition's fp32 partials and
    every padding lane of the wide kernels.  Also: the per-launch byte model sums to mdie_cdan_algorithmic_bytes (bench.py reports a
    mismatch instead of asserting it)."""
    from oracle import params as P
    sd = P.make_state_dict(42)
    x, _ = P.lowlight_batch(33, *shape)
    x = x.cuda()
    eng = E.CdanEngine("cuda", prec).load(sd)
    y0 = eng.forward(x, out=torch.empty_like(x)).clone()
    outs = []
    for fill in (0xFF, 0x7F, 0x00):
        eng._ws.fill_(fill)
        outs.append(eng.forward(x, out=torch.empty_like(x)).clone())
    torch.cuda.synchronize()
    assert torch.isfinite(y0).all()
    for fill, y in zip((0xFF, 0x7F, 0x00), outs):
        assert torch.isfinite(y).all(), f"workspace pre-filled with {fill:#x}: non-finite output"
        assert torch.equal(y, y0), f"workspace pre-filled with {fill:#x}: the output changed"
    from mdie_amd import lib as LL
    _, ex = eng.forward(x, profile=True)
    esz = 4 if prec == "fp32" else 2
    alg = LL.lib.mdie_cdan_algorithmic_bytes(shape[0], shape[1], shape[2], esz)
    assert abs(sum(b for _, b, _ in ex["launch_info"]) - alg) <= 1e-6 * alg


@pytest.mark.parametrize("prec", ["bf16", "fp16"])
@pytest.mark.parametrize("cin_segs,shape,act", [([128], (3, 32, 32), "none"), ([128, 16], (2, 40, 40), "none"), ([256, 16, 16], (5, 32, 32), "none"),
                                                ([256, 16, 16, 16], (2, 27, 21), "relu"), ([128, 16, 16, 16, 16], (1, 48, 33), "none"),
                                                ([256, 128, 64, 48], (2, 16, 24), "none")])
def test_ksplit_conv_kernel_deep_dense_layers(E, L, prec, cin_segs, shape, act):
    """conv_ksplit_kernel (csrc/conv_ksplit.hip): the 3x3 / 16-output layers of encoder.dense2 / dense3 (models/cdan.py:41-46: BN ->
    ReLU -> Conv3x3 over the concatenation of up to five segments) with the K axis split over the four waves of a workgroup --
    4..16 K chunks (1..4 per wave, a half-empty last chunk), ragged 8x8 tile edges, segments with their own strides, output into a
    slice of a wider buffer; against torch's CPU convolution on the same rounded operands.  The kernel is chosen by (layer, map)
    alone: every image of a batch reproduces its single-image run bit for bit."""
    import ctypes as C
    import torch.nn.functional as F
    dt, td = E.dtype_id(prec), TORCH_DT[prec]
    rnd = (lambda t: t.to(td).float())
    B, H, W = shape
    cin, cout = sum(cin_segs), 16
    g = torch.Generator().manual_seed(cin * 3 + H)
    strides = [c + (16 if i % 2 else 0) for i, c in enumerate(cin_segs)]
    bufs = [rnd(torch.randn(B, H, W, st, generator=g)) for st in strides]
    w = rnd(torch.randn(cout, cin, 3, 3, generator=g) * 0.05)
    sc, sh = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    ps, pt = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.3
    x = torch.cat([b[..., :c] for b, c in zip(bufs, cin_segs)], 3)
    xa = rnd(torch.relu(x * ps + pt))
    ref = F.conv2d(xa.permute(0, 3, 1, 2), w, padding=1) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    ref = (torch.relu(ref) if act == "relu" else ref).permute(0, 2, 3, 1)

    dbufs = [b.cuda().to(td) for b in bufs]
    wp = E.pack_conv_weight(w, dt, cin_stored=cin).cuda()
    dsc, dsh, dps, dpt = sc.cuda(), sh.cuda(), ps.cuda(), pt.cuda()

    def run(first, nb):
        out = torch.full((nb, H, W, 48), -7.0, device="cuda", dtype=td)
        d = L.ConvDesc()
        d.dtype, d.B, d.H, d.W, d.ksize, d.nseg = dt, nb, H, W, 3, len(dbufs)
        for i, (b, c) in enumerate(zip(dbufs, cin_segs)):
            d.inp[i] = L.Seg(b[first:].data_ptr(), c, b.shape[3])
        d.cin, d.cout = cin, cout
        d.pre_scale, d.pre_shift = dps.data_ptr(), dpt.data_ptr()
        d.weight, d.post_scale, d.post_shift = wp.data_ptr(), dsc.data_ptr(), dsh.data_ptr()
        d.act, d.pool = (L.ACT_RELU if act == "relu" else L.ACT_NONE), 0
        d.out, d.out_stride = out[..., 16:].data_ptr(), 48
        L.check(L.lib.mdie_conv_fwd(C.byref(d), None), "mdie_conv_fwd")
        torch.cuda.synchronize()
        return out

    out = run(0, B)
    assert rel_to_max(out[..., 16:32], ref) <= {"bf16": 8e-3, "fp16": 1e-3}[prec]
    assert (out[..., :16] == -7.0).all() and (out[..., 32:] == -7.0).all()     # nothing written outside the slice
    for i in range(B):                                                          # an image's bits do not depend on its batch
        assert torch.equal(run(i, 1)[0], out[i])


@pytest.mark.parametrize("prec", ["bf16", "fp16"])
@pytest.mark.parametrize("case", ["plain", "pool", "residual", "stats", "convT_ragged_batch",
                                  "plain_noact", "plain_noact_half", "convT_noact", "convT_noact_half"])
def test_wide_conv_lds_dma_kernel(E, L, prec, case):
    """conv_wide_kernel (csrc/conv_wide.hip: 32x16-pixel tiles, LDS-DMA double buffering, persistent workgroups) -- the
    kernel behind encoder.conv2-4 and decoder.conv1-3 at BASELINE sizes -- against torch's CPU convolution on the same
    rounded operands, and bit for bit against conv_kernel (which a single image, too few work items for the persistent
    grid, still runs on): border tiles, several items per workgroup, 2 and 4 K chunks, every epilogue it has.  The
    `*_noact` cases are the activation-free instantiations (conv_wide.hip: `MDIE_ACT_NONE`, full-width items and the
    half-width form taken below 192 items) that the training step selects for the forward and the input gradient of the wide
    layers at BASELINE configs[2] sizes (the convolutions under `scaler.scale(loss).backward()`, models/model.py:160-164)."""
    import ctypes as C
    import torch.nn.functional as F
    dt, td = E.dtype_id(prec), TORCH_DT[prec]
    rnd = lambda t: t.to(td).float()
    B, H, W, cin, cout = {"plain": (12, 32, 64, 64, 128), "pool": (12, 32, 64, 128, 128), "residual": (12, 32, 32, 128, 256),
                          "stats": (16, 32, 32, 64, 512), "convT_ragged_batch": (13, 16, 32, 64, 512),
                          "plain_noact": (12, 32, 64, 64, 256), "plain_noact_half": (12, 32, 64, 128, 128),
                          "convT_noact": (25, 16, 32, 128, 512), "convT_noact_half": (13, 16, 32, 64, 512)}[case]
    items = B * (H // 16) * (W // 32) * (cout // 64)
    assert items >= 96 > (H // 16) * (W // 32) * (cout // 64)    # the batch runs conv_wide, one image conv_kernel
    noact = "noact" in case
    if noact:                                                     # half-width items below 3/4 of the 256 CUs (conv_wide.hip: launch_conv_wide)
        assert (items * 4 < 256 * 3) == case.endswith("_half")
    g = torch.Generator().manual_seed(len(case) * 13 + cin)
    x = rnd(torch.randn(B, H, W, cin, generator=g))
    transposed = case.startswith("convT")
    w = rnd(torch.randn((cin, cout, 3, 3) if transposed else (cout, cin, 3, 3), generator=g) * 0.05)
    sc, sh = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    res = rnd(torch.randn(B, H, W, cout, generator=g)) if case == "residual" else None
    xin = x.permute(0, 3, 1, 2)
    ref = F.conv_transpose2d(xin, w, padding=1) if transposed else F.conv2d(xin, w, padding=1)
    ref = ref * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    if not noact:
        ref = torch.relu(ref)
    if case == "pool":
        ref = F.max_pool2d(ref, 2, 2)
    if res is not None:
        ref = ref + res.permute(0, 3, 1, 2)
    ref = ref.permute(0, 2, 3, 1)

    wp = E.pack_conv_weight(w, dt, transposed=transposed).cuda()
    dx, dsc, dsh = x.cuda().to(td), sc.cuda(), sh.cuda()
    dres = res.cuda().to(td) if res is not None else None

    def run(nb):
        Ho, Wo = (H // 2, W // 2) if case == "pool" else (H, W)
        out = torch.full((nb, Ho, Wo, cout), -7.0, device="cuda", dtype=td)
        d = L.ConvDesc()
        d.dtype, d.B, d.H, d.W, d.ksize, d.nseg = dt, nb, H, W, 3, 1
        d.inp[0] = L.Seg(dx.data_ptr(), cin, cin)
        d.cin, d.cout = cin, cout
        d.weight, d.post_scale, d.post_shift = wp.data_ptr(), dsc.data_ptr(), dsh.data_ptr()
        d.act, d.pool = (L.ACT_NONE if noact else L.ACT_RELU), int(case == "pool")
        if dres is not None:
            d.residual, d.res_stride = dres.data_ptr(), cout
        d.out, d.out_stride = out.data_ptr(), cout
        part = None
        if case == "stats":
            assert L.lib.mdie_conv_tile(nb, H, W, cout) == 16
            part = torch.full((nb, (H // 16) * (W // 16), 2, cout), 3.0, device="cuda")
            d.pool_partial = part.data_ptr()
        L.check(L.lib.mdie_conv_fwd(C.byref(d), None), "mdie_conv_fwd")
        torch.cuda.synchronize()
        return out, part

    out, part = run(B)                     # >= 96 work items: conv_wide_kernel
    assert rel_to_max(out, ref) <= {"bf16": 8e-3, "fp16": 1e-3}[prec]
    if case != "stats":
        one, _ = run(1)                    # 1 image: conv_kernel
        assert torch.equal(one[0], out[0]), "the two convolution kernels must agree bit for bit"
    else:
        o = out.float().reshape(B, H // 16, 16, W // 16, 16, cout).permute(0, 1, 3, 2, 4, 5).reshape(B, -1, 256, cout)
        assert torch.allclose(part[:, :, 0], o.sum(2), rtol=1e-5, atol=1e-3)
        assert torch.equal(part[:, :, 1], o.amax(2))


@pytest.mark.parametrize("prec", ["bf16", "fp16"])
@pytest.mark.parametrize("cin_segs,act", [([8, 16], "none"), ([8, 16, 16], "none"), ([8, 16, 16, 16], "none"), ([16, 16], "relu"), ([16, 16, 16, 8], "none"),
                                          ([24, 16], "none"), ([8, 8, 8, 8, 8], "none"), ([32, 24], "relu")])
def test_thin_persistent_conv_kernel(E, L, prec, cin_segs, act):
    """conv_thin_kernel (csrc/conv_thin.hip: persistent workgroups, one wave per 16-byte input column, next tile prefetched
    into registers, all K chunks of a tile in LDS at once) -- the kernel behind decoder.final_dense layers 1..3 at BASELINE
    sizes (base + 1..3 growth maps = 3, 5, 7 live columns) -- against torch's CPU convolution on the same rounded operands,
    and bit for bit against conv_kernel, which a batch with fewer than 1024 tiles still runs on: every image border, runs of
    several tiles per workgroup that cross image boundaries, one and two K chunks, segments with their own strides, the
    output written into a slice of a wider buffer.  Both staging forms: the column form (one 16-byte column per wave: two
    units, or more than four -- [8]*5) and the PAIR form (a wave stages a <= 16-channel stretch of one segment with lane =
    (pixel, half): three or four units, single- and two-column units mixed, a 24-channel segment cut into 16 + 8)."""
    import ctypes as C
    import torch.nn.functional as F
    dt, td = E.dtype_id(prec), TORCH_DT[prec]
    rnd = lambda t: t.to(td).float()
    B, H, W = 5, 240, 256                  # 5 * 15 * 16 = 1200 tiles: conv_thin; 2 images = 480 tiles: conv_kernel
    cin, cout = sum(cin_segs), 16
    g = torch.Generator().manual_seed(cin * 11 + len(cin_segs))
    strides = [c + (8 if i % 2 else 0) for i, c in enumerate(cin_segs)]
    bufs = [rnd(torch.randn(B, H, W, st, generator=g)) for st in strides]
    w = rnd(torch.randn(cout, cin, 3, 3, generator=g) * 0.2)
    sc, sh = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    ps, pt = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.3
    x = torch.cat([b[..., :c] for b, c in zip(bufs, cin_segs)], 3)
    xa = rnd(torch.relu(x * ps + pt))
    ref = (F.conv2d(xa.permute(0, 3, 1, 2), w, padding=1) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)).permute(0, 2, 3, 1)
    if act == "relu":
        ref = torch.relu(ref)

    dbufs = [b.cuda().to(td) for b in bufs]
    wp = E.pack_conv_weight(w, dt, cin_stored=cin).cuda()
    dsc, dsh, dps, dpt = sc.cuda(), sh.cuda(), ps.cuda(), pt.cuda()

    def run(nb):
        out = torch.full((nb, H, W, 40), -7.0, device="cuda", dtype=td)
        d = L.ConvDesc()
        d.dtype, d.B, d.H, d.W, d.ksize, d.nseg = dt, nb, H, W, 3, len(dbufs)
        for i, (b, c) in enumerate(zip(dbufs, cin_segs)):
            d.inp[i] = L.Seg(b.data_ptr(), c, b.shape[3])
        d.cin, d.cout = cin, cout
        d.pre_scale, d.pre_shift = dps.data_ptr(), dpt.data_ptr()
        d.weight, d.post_scale, d.post_shift = wp.data_ptr(), dsc.data_ptr(), dsh.data_ptr()
        d.act, d.pool = (L.ACT_RELU if act == "relu" else L.ACT_NONE), 0
        d.out, d.out_stride = out[..., 16:].data_ptr(), 40
        L.check(L.lib.mdie_conv_fwd(C.byref(d), None), "mdie_conv_fwd")
        torch.cuda.synchronize()
        return out

    out = run(B)
    assert rel_to_max(out[..., 16:32], ref) <= {"bf16": 8e-3, "fp16": 1e-3}[prec]
    assert (out[..., :16] == -7.0).all() and (out[..., 32:] == -7.0).all()     # nothing written outside the slice
    two = run(2)
    assert torch.equal(two, out[:2]), "the two convolution kernels must agree bit for bit"


@pytest.mark.parametrize("prec", ["bf16", "fp16"])
@pytest.mark.parametrize("cin_segs,hw", [([256], (32, 32)), ([256, 16, 16, 16], (32, 32)), ([128, 16], (24, 40)), ([256, 16], (8, 8))])
def test_small_map_deep_k_conv_tile8_equals_tile16(E, L, prec, cin_segs, hw):
    """The DenseLayers on small, deep maps (encoder.dense3 / dense2 at small batches, models/cdan.py:64-65: 16 outputs from
    128-304 channels) run conv_kernel with 8x8 tiles; inside a batch large enough to fill the chip they run 16x16 tiles.  Both
    against torch's CPU convolution of the same rounded operands (ragged tiles, several segments, K not a multiple of the
    chunk) and against each other BIT FOR BIT -- the kernel choice follows the batch, an image's bits must not."""
    import ctypes as C
    import torch.nn.functional as F
    dt, td = E.dtype_id(prec), TORCH_DT[prec]
    rnd = lambda t: t.to(td).float()
    H, W = hw
    cin, cout = sum(cin_segs), 16
    tiles16 = -(-H // 16) * -(-W // 16)
    Bbig = -(-512 // tiles16)                  # >= 512 workgroups of 16x16 tiles: the TILE = 16 kernel
    Bsmall = 3
    assert Bsmall * tiles16 < 512 and cin >= 128
    g = torch.Generator().manual_seed(cin + H)
    bufs = [rnd(torch.randn(Bbig, H, W, c, generator=g)) for c in cin_segs]
    w = rnd(torch.randn(cout, cin, 3, 3, generator=g) * 0.05)
    sh = torch.randn(cout, generator=g) * 0.1
    ps, pt = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.3
    x = torch.cat([b[:Bsmall] for b in bufs], 3)
    ref = (F.conv2d(rnd(torch.relu(x * ps + pt)).permute(0, 3, 1, 2), w, padding=1) + sh.view(1, -1, 1, 1)).permute(0, 2, 3, 1)
    dbufs = [b.cuda().to(td) for b in bufs]
    wp = E.pack_conv_weight(w, dt, cin_stored=cin).cuda()
    dsc, dsh, dps, dpt = torch.ones(cout).cuda(), sh.cuda(), ps.cuda(), pt.cuda()

    def run(nb):
        out = torch.full((nb, H, W, 16), -7.0, device="cuda", dtype=td)
        d = L.ConvDesc()
        d.dtype, d.B, d.H, d.W, d.ksize, d.nseg = dt, nb, H, W, 3, len(dbufs)
        for i, (b, c) in enumerate(zip(dbufs, cin_segs)):
            d.inp[i] = L.Seg(b.data_ptr(), c, c)
        d.cin, d.cout = cin, cout
        d.pre_scale, d.pre_shift = dps.data_ptr(), dpt.data_ptr()
        d.weight, d.post_scale, d.post_shift = wp.data_ptr(), dsc.data_ptr(), dsh.data_ptr()
        d.act, d.pool = L.ACT_NONE, 0
        d.out, d.out_stride = out.data_ptr(), 16
        L.check(L.lib.mdie_conv_fwd(C.byref(d), None), "mdie_conv_fwd")
        torch.cuda.synchronize()
        return out

    small = run(Bsmall)
    assert rel_to_max(small, ref) <= {"bf16": 8e-3, "fp16": 1e-3}[prec]
    big = run(Bbig)
    assert torch.equal(big[:Bsmall], small), "the 8x8-tile kernel must agree with the 16x16-tile kernel bit for bit"


# ---------------------------------------------------------------------------------------------------------------------
# training mode: HIP convolutions (forward / dgrad / wgrad) under autograd
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("prec", PRECISIONS)
@pytest.mark.parametrize("transposed", [False, True])
@pytest.mark.parametrize("ks,cin_segs,cout,hw", [(3, [16], 16, (8, 12)), (3, [32, 16], 64, (16, 16)), (1, [64, 16, 16], 64, (8, 8)),
                                                 (3, [64], 128, (5, 7)), (3, [16], 64, (20, 37)), (3, [64, 16, 16, 16], 16, (33, 18)),
                                                 (1, [16, 16, 16, 16, 16], 16, (17, 40)), (3, [128], 64, (32, 48))])
def test_conv_autograd_against_torch_cpu(E, L, ks, cin_segs, cout, hw, transposed, prec):
    """forward, input gradient and weight/bias gradients of the conv Function vs torch's CPU convolution.
    bf16: operands are rounded to bf16 first, so products are exact on both sides and only the bf16 rounding of the
    stored forward output / input gradient (2^-8 relative) and fp32 summation order differ."""
    import torch.nn.functional as F
    import mdie_amd.train as T
    if transposed and ks == 1:
        pytest.skip("the network has no 1x1 transposed convolution")
    bf = prec != "fp32"
    td = TORCH_DT[prec]
    rnd = (lambda t: t.to(td).float())
    g = torch.Generator().manual_seed(ks * 100 + cout)
    cin = sum(cin_segs)
    segs = [rnd(torch.randn(2, c, *hw, generator=g)) for c in cin_segs]
    w = rnd(torch.randn((cin, cout, ks, ks) if transposed else (cout, cin, ks, ks), generator=g) * 0.1)
    b = torch.randn(cout, generator=g)
    dy = rnd(torch.randn(2, cout, *hw, generator=g))
    # reference on the CPU
    rs = [s.clone().requires_grad_(True) for s in segs]
    rw, rb = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    x = torch.cat(rs, 1)
    ry = F.conv_transpose2d(x, rw, rb, padding=ks // 2) if transposed else F.conv2d(x, rw, rb, padding=ks // 2)
    ry.backward(dy)
    # engine
    gs = [s.cuda().to(td).contiguous(memory_format=torch.channels_last).requires_grad_(True) for s in segs]
    gw, gb = w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    y = T.conv(E.dtype_id(prec), gw, gb, gs, transposed=transposed)
    y.backward(dy.cuda().to(td))
    act_tol = {"fp32": 2e-5, "bf16": 8e-3, "fp16": 1e-3}[prec]      # 16-bit: one rounding of the stored result (2^-8 / 2^-11 relative)
    assert rel_to_max(y, ry) <= act_tol
    assert rel_to_max(gw.grad, rw.grad) <= 5e-5
    assert rel_to_max(gb.grad, rb.grad) <= 5e-5
    for a, r in zip(gs, rs):
        assert rel_to_max(a.grad, r.grad) <= (act_tol if bf else 5e-5)


def test_train_step_matches_reference(E, golden_dir):
    """One training-mode forward/backward (batch-stat BatchNorm, dropout off as in the fixture) against the values the
    reference produced: output, charbonnier loss, twelve parameter gradients, updated running statistics."""
    import json as _json
    from models.cdan import CDAN
    from oracle import params as P
    z = np.load(os.path.join(golden_dir, "train_step_32.npz"))
    net = CDAN(precision="fp32")
    net.load_state_dict(P.make_state_dict(42), strict=True)
    net = net.cuda().train()
    net.dropout_p = 0.0
    y = net(torch.from_numpy(z["x"]).cuda())
    t = torch.from_numpy(z["t"]).cuda()
    loss = torch.sqrt((y - t) ** 2 + 1e-6).mean()
    loss.backward()
    assert rel_to_max(y, torch.from_numpy(z["y"])) <= 1e-4
    assert loss.item() == pytest.approx(float(z["loss"]), rel=1e-5)
    named = dict(net.named_parameters())
    # Gradients pass through 32 batch-statistic BatchNorms over as few as 32 samples (2 x 4 x 4 at the bottleneck):
    # fp32 summation-order noise is amplified towards the first layers, hence 5.5e-3 here (the CPU oracle, which
    # shares ATen's summation order with the reference, is held to 2e-4 in tests/test_oracle_golden.py).
    # (a bias that feeds a batch-statistic BatchNorm has an exactly-zero true gradient: both sides hold rounding noise)
    errs = {k[2:]: rel_to_max(named[k[2:]].grad, torch.from_numpy(z[k])) for k in z.files
            if k.startswith("g:") and np.abs(z[k]).max() > 1e-7}
    print({k: f"{v:.1e}" for k, v in errs.items()})
    assert max(errs.values()) <= 5.5e-3, errs      # measured 1.8e-3 (x3)
    sd = net.state_dict()
    for k in z.files:
        if k.startswith("s:"):
            assert rel_to_max(sd[k[2:]], torch.from_numpy(z[k])) <= 1e-4, k
    norms = _json.loads(str(z["grad_norms"]))
    bad = [k for k, n in norms.items() if abs(float(named[k].grad.double().norm()) - n) > 1e-2 * n + 1e-6]
    assert not bad, bad[:5]
    assert int(sd["encoder.conv1.bn.num_batches_tracked"]) == 1


def test_ddp_two_shard_gradients_match_reference(E, golden_dir):
    """Data parallel, N = 2 (BASELINE configs[2]): each rank runs the HIP training graph on its
    shard; the exchanged gradient is the mean over ranks.  tests/golden/ddp_2shard_32.npz holds what the REFERENCE's step
    (models/model.py:159-164) gives per shard; here both shards go through the HIP graph, the per-shard gradients are
    averaged the way GradBuckets.finish() does (flat fp32 bucket: sum, then / world) and compared with the mean of the
    reference's.  (tests/test_bench_sharding_cpu.py pushes the same fixture through the real GradBuckets over gloo.)"""
    from models.cdan import CDAN
    from oracle import params as P
    z = np.load(os.path.join(golden_dir, "ddp_2shard_32.npz"))
    x, t = torch.from_numpy(z["x"]).cuda(), torch.from_numpy(z["t"]).cuda()
    keys = [k[3:] for k in z.files if k.startswith("g0:")]
    per = []
    for r, sl in enumerate((slice(0, 2), slice(2, 4))):
        net = CDAN(precision="fp32")
        net.load_state_dict(P.make_state_dict(42), strict=True)       # identical replicas
        net = net.cuda().train()
        net.dropout_p = 0.0
        y = net(x[sl])
        loss = torch.sqrt((y - t[sl]) ** 2 + 1e-6).mean()
        loss.backward()
        assert rel_to_max(y, torch.from_numpy(z[f"y{r}"])) <= 1e-4
        assert loss.item() == pytest.approx(float(z[f"loss{r}"]), rel=1e-5)
        named = dict(net.named_parameters())
        per.append({k: named[k].grad.detach().clone() for k in keys})
        errs = {k: rel_to_max(per[r][k], torch.from_numpy(z[f"g{r}:" + k])) for k in keys if np.abs(z[f"g{r}:" + k]).max() > 1e-7}
        assert max(errs.values()) <= 5.5e-3, (r, errs)
        if r == 0:
            norms = __import__("json").loads(str(z["norms0"]))
            bad = [k for k, n in norms.items() if abs(float(named[k].grad.double().norm()) - n) > 1e-2 * n + 1e-6]
            assert not bad, bad[:5]
    flat = torch.cat([per[0][k].reshape(-1) for k in keys]) + torch.cat([per[1][k].reshape(-1) for k in keys])   # all-reduce(SUM)
    flat /= 2                                                                                                      # / world
    off = 0
    for k in keys:
        n = per[0][k].numel()
        mean_ref = (torch.from_numpy(z["g0:" + k]).double() + torch.from_numpy(z["g1:" + k]).double()) / 2
        if float(mean_ref.abs().max()) > 1e-7:
            assert rel_to_max(flat[off:off + n].view_as(per[0][k]), mean_ref.float()) <= 5.5e-3, k
        off += n


def test_training_reduces_loss_and_dropout_is_active(E):
    from models.cdan import CDAN
    from oracle import params as P
    torch.manual_seed(0)
    net = CDAN(precision="bf16").cuda().train()
    x, t = P.lowlight_batch(31, 4, 32, 32)
    x, t = x.cuda(), t.cuda()
    with torch.no_grad():
        a, b = net(x), net(x)
    assert not torch.equal(a, b)               # dropout masks differ between calls
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    losses = []
    for _ in range(12):
        opt.zero_grad()
        loss = torch.sqrt((net(x) - t) ** 2 + 1e-6).mean()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert all(np.isfinite(losses)) and min(losses[-3:]) < losses[0]


def test_run_py_train_phase(E, tmp_path):
    from PIL import Image
    from oracle import params as P
    from mdie_amd import host as H
    root = str(tmp_path)
    _, clean = P.lowlight_batch(5, 4, 32, 32)
    deg = (clean * 0.3)
    for sub, data in (("degraded", deg), ("clean", clean)):
        os.makedirs(os.path.join(root, "data", sub))
        for i in range(4):
            Image.fromarray((data[i].permute(1, 2, 0).numpy() * 255).round().astype(np.uint8)).save(os.path.join(root, "data", sub, f"{i}.png"))
    cfg = H.load_config(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "config", "example_noise_64.json"), "train")
    cfg["loss"] = H._wrap({"enabled": True, "terms": [{"name": "charbonnier", "weight": 1.0}, {"name": "ssim", "weight": 0.5}]})
    cfg["train"]["dataset"] = H._wrap({"name": ["data.dataset", "PairedDataset"], "args": {
        "input_root": os.path.join(root, "data", "degraded"), "target_root": os.path.join(root, "data", "clean"), "pairing_mode": "filename",
        "transform": {"backend": "albumentations", "ops": [{"name": "HorizontalFlip", "args": {"p": 0.5}},
                                                           {"name": "RandomBrightnessContrast", "args": {"brightness_limit": 0.1, "contrast_limit": 0.1, "p": 0.25}},
                                                           {"name": "RandomGamma", "args": {"gamma_limit": [70, 130], "p": 0.2}},     # config/low_light.json:102-103
                                                           {"name": "Normalize", "args": {"mean": [0, 0, 0], "std": [1, 1, 1]}},
                                                           {"name": "ToTensorV2", "args": {}}]}}})
    cfg["train"]["dataloader"] = H._wrap({"args": {"batch_size": 2, "shuffle": True, "num_workers": 0}})
    cfg["train"]["n_epoch"], cfg["train"]["model_path"], cfg["train"]["model_name"] = 2, os.path.join(root, "weights"), "CDAN_t.pt"
    cfg["logging"]["root_dir"] = os.path.join(root, "runs")
    model = H.run(cfg)
    assert len(model.history) == 2 and all(np.isfinite(h["total"]) for h in model.history)
    assert set(model.history[0]) == {"total", "charbonnier", "ssim"}
    assert len(torch.load(os.path.join(root, "weights", "CDAN_t.pt"))) == 236


# ---------------------------------------------------------------------------------------------------------------------
# training loss on the device: value and gradient of every network-free term vs the CPU oracle
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("terms,shape", [
    ([("charbonnier", 1.0, 1e-3), ("ssim", 0.5, 0.0)], (2, 3, 64, 64)),                                   # blur / noise configs
    ([("mse", 1.0, 0.0), ("ssim", 0.5, 0.0)], (1, 3, 40, 56)),                                            # low_light (network-free part)
    ([("l1", 1.0, 0.0), ("ssim", 0.5, 0.0)], (3, 3, 23, 37)),                                             # jpeg, ragged tiles
    ([("charbonnier", 1.0, 1e-3), ("ssim", 0.5, 0.0), ("gradient_l1", 0.35, 1.0)], (2, 3, 48, 33)),       # pixelation_hard
    ([("gradient_l1", 1.0, 0.0), ("mse", 0.25, 0.0)], (2, 3, 17, 16)),
    ([("ssim", 1.0, 0.0)], (1, 3, 11, 11)),                                                               # one window
])
def test_fused_loss_matches_oracle(E, terms, shape):
    from mdie_amd import pipeline as PL
    from oracle import loss_oracle as LO
    g = torch.Generator().manual_seed(shape[2] * 31 + shape[3])
    t = torch.rand(*shape, generator=g)
    o = (t * 0.6 + 0.15 * torch.rand(*shape, generator=g)).clamp(0, 1)
    ro = o.double().requires_grad_(True)
    rtotal, rvals = LO.pipeline(ro, t.double(), terms)
    rtotal.backward()
    go = o.cuda().requires_grad_(True)
    total, values = PL.fused_loss(go, t.cuda(), terms)
    (2.0 * total).backward()                               # upstream gradient is honoured
    assert total.item() == pytest.approx(rtotal.item(), rel=2e-5)           # fp32 kernels vs fp64 oracle
    for k, rv in enumerate(rvals):
        assert values[k].item() == pytest.approx(rv.item(), rel=2e-5, abs=1e-7), terms[k][0]
    assert values[-1].item() == pytest.approx(rtotal.item(), rel=2e-5)
    # sign() terms flip where |d| is at rounding level: compare where the fp64 argument is clear of zero
    err = (go.grad.cpu().double() / 2.0 - ro.grad).abs()
    tol = 2e-4 * ro.grad.abs().max().item()
    if any(n in ("l1", "gradient_l1") for n, _, _ in terms):
        assert (err > tol).double().mean().item() < 1e-3
    else:
        assert err.max().item() <= tol


def test_fused_loss_values_only_and_errors(E):
    from mdie_amd import pipeline as PL
    t = torch.rand(1, 3, 16, 16).cuda()
    o = torch.rand(1, 3, 16, 16).cuda()
    total, values = PL.fused_loss(o, t, [("mse", 1.0, 0.0)])          # no grad requested: values only
    assert not total.requires_grad and total.item() == pytest.approx(((o - t) ** 2).mean().item(), rel=1e-5)
    with pytest.raises(Exception):
        PL.fused_loss(o.cpu(), t.cpu(), [("mse", 1.0, 0.0)])           # no CPU fallback
    with pytest.raises(Exception):
        PL.fused_loss(o[:, :, :8, :8], t[:, :, :8, :8], [("ssim", 1.0, 0.0)])   # smaller than one 11x11 window
    with pytest.raises(Exception):
        PL.fused_loss(o, t, [("mse", 1.0, 0.0), ("mse", 1.0, 0.0)])


# ---------------------------------------------------------------------------------------------------------------------
# native training blocks (csrc/bn.hip + conv fwd/dgrad/wgrad) vs the CPU oracle differentiated by autograd, fp32 vs fp64
# ---------------------------------------------------------------------------------------------------------------------
_SD_CACHE = {}


def _train_net(seed=42):
    from models.cdan import CDAN
    from oracle import params as P
    net = CDAN(precision="fp32")
    if seed not in _SD_CACHE:
        _SD_CACHE[seed] = P.make_state_dict(seed)
    sd = {k: v.clone() for k, v in _SD_CACHE[seed].items()}
    net.load_state_dict(sd, strict=True)
    return net.cuda().train(), {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}


def _leaf(sd, keys):
    for k in keys:
        sd[k] = sd[k].clone().requires_grad_(True)


def _nhwc_cuda(t, pad_to=None):
    t = t.float()
    if pad_to and t.shape[1] < pad_to:
        t = torch.nn.functional.pad(t, (0, 0, 0, 0, 0, pad_to - t.shape[1]))
    return t.cuda().contiguous(memory_format=torch.channels_last)


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_dense_block_gradients_with_fp32_running_sums(E, L, precision, monkeypatch):
    """MDIE_TRAIN_ACC32 path of bn_bwd_apply_kernel (segment gradients summed in fp32 over the consuming layers, rounded
    once): same DenseBlock backward as the default path up to the roundings it removes, and closer to (never further
    from) the fp32 engine's gradients."""
    import mdie_amd.train as T
    net, _ = _train_net()
    blk = net.encoder.dense1
    g = torch.Generator().manual_seed(4)
    x32 = torch.randn(2, 64, 12, 20, generator=g)
    dy32 = torch.randn(2, 64, 12, 20, generator=g)

    def run(prec, acc32):
        monkeypatch.setattr(T, "ACC32", acc32)
        td = TORCH_DT[prec]
        gx = x32.cuda().to(td).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        net.zero_grad(set_to_none=True)
        y = T.dense_block(E.dtype_id(prec), blk, gx, 64)
        y.backward(dy32.cuda().to(td).contiguous(memory_format=torch.channels_last))
        return gx.grad.float(), [p.grad.clone() for p in blk.parameters()]

    ref_dx, ref_dp = run("fp32", False)
    dx0, dp0 = run(precision, False)
    dx1, dp1 = run(precision, True)
    tol = {"bf16": 3e-2, "fp16": 4e-3}[precision]          # measured 4.6e-3 / 6.1e-4: the roundings the fp32 sums remove
    assert rel_to_max(dx1, dx0) <= tol
    # against the fp32 engine both differ mostly through the 16-bit FORWARD (batch statistics, ReLU masks): compare in L2,
    # and require that rounding once is not worse than rounding five times
    l2 = lambda a: float((a - ref_dx).double().norm() / ref_dx.double().norm())
    assert l2(dx1) <= 1.02 * l2(dx0) + 1e-6
    for a, b in zip(dp1, dp0):
        assert rel_to_max(a, b) <= tol


@pytest.mark.parametrize("stage,hw", [(2, (12, 8)), (3, (6, 10)), (4, (5, 7))])
def test_conv_block_train_matches_oracle(E, L, stage, hw):
    """conv -> batch-stat BN -> ReLU (-> maxpool): outputs, running statistics and every gradient."""
    import mdie_amd.train as T
    from oracle import cdan_oracle as O
    net, sd = _train_net()
    blk = getattr(net.encoder, f"conv{stage}")
    p = f"encoder.conv{stage}"
    keys = [p + ".conv.weight", p + ".conv.bias", p + ".bn.weight", p + ".bn.bias"]
    _leaf(sd, keys)
    pool = stage < 4
    g = torch.Generator().manual_seed(stage)
    cin = blk.conv.weight.shape[1]
    x = torch.randn(2, cin, *hw, generator=g, dtype=torch.float64)
    rx = x.clone().requires_grad_(True)
    stats = {}
    ro = O.conv_block(sd, p, rx, "train", stats)
    if pool:
        ro = torch.nn.functional.max_pool2d(ro, 2, 2)
    gy = torch.randn(ro.shape, generator=g, dtype=torch.float64)
    ro.backward(gy)
    gx = _nhwc_cuda(x).requires_grad_(True)
    o = T.conv_block(L.F32, blk, gx, pool, 0.0, need_t=False)
    o.backward(_nhwc_cuda(gy))
    assert rel_to_max(o, ro) <= 2e-5
    assert rel_to_max(gx.grad, rx.grad) <= 2e-4
    assert rel_to_max(blk.conv.weight.grad, sd[keys[0]].grad) <= 2e-4
    assert blk.conv.bias.grad.abs().max().item() == 0.0 and sd[keys[1]].grad.abs().max().item() < 1e-9 * gy.abs().sum().item()
    assert rel_to_max(blk.bn.weight.grad, sd[keys[2]].grad) <= 2e-4
    assert rel_to_max(blk.bn.bias.grad, sd[keys[3]].grad) <= 2e-4
    assert rel_to_max(blk.bn.running_mean, stats[p + ".bn.running_mean"]) <= 2e-5
    assert rel_to_max(blk.bn.running_var, stats[p + ".bn.running_var"]) <= 2e-5


def test_conv_block_dropout_mask_and_gradient(E, L):
    """dropout: out_t = out_o * mask / (1 - p) with ~p of the elements dropped, the same mask in backward, a new one per call."""
    import mdie_amd.train as T
    net, _ = _train_net()
    blk = net.encoder.conv2
    torch.manual_seed(5)
    x = _nhwc_cuda(torch.randn(2, 64, 16, 16)).requires_grad_(True)
    o, t = T.conv_block(L.F32, blk, x, True, 0.2)
    live = o > 0
    ratio = (t[live] / o[live])
    assert set(torch.unique(ratio.round(decimals=4)).tolist()) == {0.0, 1.25}
    assert 0.15 < (ratio == 0).float().mean().item() < 0.25
    o2, t2 = T.conv_block(L.F32, blk, x, True, 0.2)
    assert torch.equal(o, o2) and not torch.equal(t, t2)
    # gradient through t only == gradient through o with the mask applied by hand
    gt = torch.randn_like(t)
    (gx_t,) = torch.autograd.grad(t, x, gt, retain_graph=True)
    (gx_o,) = torch.autograd.grad(o, x, gt * (t != 0) * 1.25 + gt * ((t == 0) & ~live) * 0.0)
    # (positions where o == 0 carry no gradient either way: ReLU)
    assert rel_to_max(gx_t, gx_o) <= 1e-5


@pytest.mark.parametrize("name,real_c,hw,sigmoid", [("encoder.dense1", 64, (10, 12), False), ("encoder.dense3", 256, (4, 6), False),
                                                    ("decoder.final_dense", 3, (8, 8), True), ("decoder.final_dense", 3, (9, 20), False)])
def test_dense_block_train_matches_oracle(E, L, name, real_c, hw, sigmoid):
    import mdie_amd.train as T
    from oracle import cdan_oracle as O
    net, sd = _train_net()
    blk = net.get_submodule(name)
    keys = [k for k in sd if k.startswith(name + ".") and sd[k].is_floating_point() and "running" not in k]
    _leaf(sd, keys)
    g = torch.Generator().manual_seed(real_c)
    x = torch.randn(2, real_c, *hw, generator=g, dtype=torch.float64)
    rx = x.clone().requires_grad_(True)
    stats = {}
    ry = O.dense_block(sd, name, rx, "train", stats)
    if sigmoid:
        ry = torch.sigmoid(ry)
    gy = torch.randn(ry.shape, generator=g, dtype=torch.float64)
    ry.backward(gy)
    gx = _nhwc_cuda(x, 16).requires_grad_(True)
    y = T.dense_block(L.F32, blk, gx, real_c, sigmoid=sigmoid)
    y.backward(gy.float().cuda() if sigmoid else _nhwc_cuda(gy, 16))
    assert rel_to_max(y[:, :ry.shape[1]], ry) <= 5e-5
    assert rel_to_max(gx.grad[:, :real_c], rx.grad) <= 5e-4
    if real_c < 16:
        assert gx.grad[:, real_c:].abs().max().item() == 0.0
    named = dict(blk.named_parameters())
    for k in keys:
        ref = sd[k].grad
        got = named[k[len(name) + 1:]].grad
        if k.endswith(".2.bias") and "transition" not in k:
            # bias of a conv whose output only ever feeds batch-statistic BatchNorms: exactly zero
            assert got.abs().max().item() == 0.0 and ref.abs().max().item() < 1e-9 * gy.abs().sum().item()
        else:
            assert rel_to_max(got, ref) <= 5e-4, k
    for k, v in stats.items():
        assert rel_to_max(net.state_dict()[k], v) <= 5e-5, k


@pytest.mark.parametrize("i,hw", [(1, (4, 6)), (2, (6, 5)), (3, (8, 8)), (4, (7, 9))])
def test_decoder_stage_train_matches_oracle(E, L, i, hw):
    """ConvTranspose -> batch-stat BN -> ReLU (-> bilinear x2) + skip."""
    import mdie_amd.train as T
    from oracle import cdan_oracle as O
    net, sd = _train_net()
    cv, bn = getattr(net.decoder, f"conv{i}"), getattr(net.decoder, f"bn{i}")
    keys = [f"decoder.conv{i}.weight", f"decoder.conv{i}.bias", f"decoder.bn{i}.weight", f"decoder.bn{i}.bias"]
    _leaf(sd, keys)
    up = i > 1
    g = torch.Generator().manual_seed(i)
    cin, cout = cv.weight.shape[0], cv.weight.shape[1]
    x = torch.randn(2, cin, *hw, generator=g, dtype=torch.float64)
    skip = torch.randn(2, cout, hw[0] * (2 if up else 1), hw[1] * (2 if up else 1), generator=g, dtype=torch.float64)
    rx, rs = x.clone().requires_grad_(True), skip.clone().requires_grad_(True)
    stats = {}
    r = O._deconv_bn_relu(sd, i, rx, "train", stats)
    r = (O.up2(r) if up else r) + rs
    gy = torch.randn(r.shape, generator=g, dtype=torch.float64)
    r.backward(gy)
    gx, gs = _nhwc_cuda(x).requires_grad_(True), _nhwc_cuda(skip, 16).requires_grad_(True)
    out = T.deconv_stage(L.F32, cv, bn, gx, gs, up)
    out.backward(_nhwc_cuda(gy, 16))
    assert rel_to_max(out[:, :cout], r) <= 2e-5
    assert rel_to_max(gx.grad, rx.grad) <= 2e-4
    assert rel_to_max(gs.grad[:, :cout], rs.grad) <= 1e-6
    assert rel_to_max(cv.weight.grad, sd[keys[0]].grad) <= 2e-4
    assert cv.bias.grad.abs().max().item() == 0.0
    assert rel_to_max(bn.weight.grad, sd[keys[2]].grad) <= 2e-4
    assert rel_to_max(bn.bias.grad, sd[keys[3]].grad) <= 2e-4
    for k, v in stats.items():
        assert rel_to_max(net.state_dict()[k], v) <= 2e-5, k


@pytest.mark.parametrize("name,C,hw,with_mul", [("bottleneck", 512, (4, 6), False), ("decoder.cbam1", 256, (8, 8), True),
                                                ("decoder.cbam2", 128, (10, 12), True), ("decoder.cbam3", 64, (24, 17), True)])
def test_cbam_train_matches_oracle(E, L, name, C, hw, with_mul):
    """CBAM with the spatial gate's BatchNorm on batch statistics (* dense_k): output, running statistics, every gradient."""
    import mdie_amd.train as T
    from oracle import cdan_oracle as O
    net, sd = _train_net()
    node = net.get_submodule(name)
    keys = [k for k in sd if k.startswith(name + ".") and sd[k].is_floating_point() and "running" not in k]
    _leaf(sd, keys)
    g = torch.Generator().manual_seed(C)
    # post-ReLU-like input: non-negative with exact zeros, so the arg-max tie rules matter
    x = torch.relu(torch.randn(2, C, *hw, generator=g, dtype=torch.float64))
    mul = torch.randn(2, C, *hw, generator=g, dtype=torch.float64) if with_mul else None
    rx = x.clone().requires_grad_(True)
    rm = mul.clone().requires_grad_(True) if with_mul else None
    stats = {}
    ry = O.cbam(sd, name, rx, "train", stats)
    if with_mul:
        ry = ry * rm
    gy = torch.randn(ry.shape, generator=g, dtype=torch.float64)
    ry.backward(gy)
    gx = _nhwc_cuda(x).requires_grad_(True)
    gm = _nhwc_cuda(mul).requires_grad_(True) if with_mul else None
    y = T.cbam(L.F32, node, gx, gm)
    y.backward(_nhwc_cuda(gy))
    assert rel_to_max(y, ry) <= 2e-5
    assert rel_to_max(gx.grad, rx.grad) <= 2e-4
    if with_mul:
        assert rel_to_max(gm.grad, rm.grad) <= 2e-5
    named = dict(node.named_parameters())
    for k in keys:
        assert rel_to_max(named[k[len(name) + 1:]].grad, sd[k].grad) <= 5e-4, k
    for k, v in stats.items():
        assert rel_to_max(net.state_dict()[k], v) <= 2e-5, k


def test_cbam_module_train_mode(E):
    """the public CBAM module (models.cbam.CBAM) in training mode: NCHW fp32 in / out, differentiable"""
    from models.cbam import CBAM
    torch.manual_seed(3)
    m = CBAM(64).cuda().train()
    x = torch.rand(2, 64, 12, 12, device="cuda", requires_grad=True)
    y = m(x)
    assert y.shape == x.shape and y.dtype == torch.float32
    y.square().mean().backward()
    assert x.grad is not None and torch.isfinite(x.grad).all()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
    assert int(m.SpatialGate.spatial.bn.num_batches_tracked) == 1


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE configs[3] (routed mixed degradations) and configs[4] (1024x1024, batch 1)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("prec,hw", [("bf16", (32, 40)), ("fp16", (64, 64)), ("fp32", (32, 40)), ("bf16", (128, 192))])
def test_routed_inference_matches_per_task_engines(E, prec, hw):
    """images labelled with a task run with that task's weights -- as ONE launch chain whose kernels look up each image's
    weight set; the result is bitwise what a dedicated engine returns for the same
    image (batch independence), in the caller's order; unlabelled images pass through"""
    from oracle import params as P
    tasks = {"noise": 11, "blur": 12, "low_light": 13}
    routed = E.RoutedEngine("cuda", prec)
    single = {}
    for t, seed in tasks.items():
        sd = P.make_state_dict(seed)
        routed.load_task(t, sd)
        single[t] = E.CdanEngine("cuda", prec).load(sd)
    x, _ = P.lowlight_batch(9, 7, *hw)
    x = x.cuda()
    labels = ["blur", "noise", "blur", "low_light", "noise", "noise", "blur"]
    y = routed.forward(x, labels)
    for i, t in enumerate(labels):
        assert torch.equal(y[i], single[t].forward(x[i:i + 1])[0]), (i, t)
    some = ["blur", None, "low_light", None, None, "noise", "blur"]
    y2 = routed.forward(x, some)
    for i, t in enumerate(some):
        assert torch.equal(y2[i], x[i] if t is None else y[i] if t == labels[i] else single[t].forward(x[i:i + 1])[0]), (i, t)
    assert torch.equal(routed.forward(x, [None] * 7), x)
    one = routed.forward(x, ["noise"] * 7)                # a single weight set through the several-sets kernels
    assert torch.equal(one, single["noise"].forward(x))
    with pytest.raises(Exception):
        routed.forward(x, ["jpeg"] * 7)
    with pytest.raises(Exception):
        routed.forward(x, labels[:3])


def test_serving_loop_matches_one_batch_at_a_time(E, net):
    """pipeline.ServingLoop (uint8 batches over PCIe, three in flight on three streams, one engine per stream) returns, in submission order,
    exactly what the one-batch-at-a-time path returns -- also when the batch shape changes mid-stream and with post-processing on"""
    from mdie_amd import pipeline as PL
    from oracle import params as P
    net.precision = "bf16"
    shapes = [(4, 64, 64)] * 4 + [(2, 32, 48)] * 2 + [(4, 64, 64)] * 3
    batches = [(P.lowlight_batch(60 + i, *s)[0].permute(0, 2, 3, 1) * 255).round().to(torch.uint8).contiguous() for i, s in enumerate(shapes)]
    for pp in (None, {"enabled": True, "ops": [{"name": "enhance_contrast", "args": {"contrast_factor": 1.2}}, {"name": "sharpen"}]}):
        with torch.no_grad():
            refs = []
            for b in batches:
                y = net(PL.feed_uint8(b.cuda()))
                refs.append((PL.apply_postprocessing(y, pp, want_uint8=True)[1] if pp else PL.to_uint8_hwc(y)).cpu())
        torch.cuda.synchronize()
        loop = PL.ServingLoop(net, depth=3, postprocessing=pp)
        n = 0
        for out, ref in zip(loop.run(batches), refs):
            assert out.shape == ref.shape and torch.equal(out, ref), f"batch {n} differs"
            n += 1
        assert n == len(batches)
    with pytest.raises(Exception):
        list(PL.ServingLoop(net).run([torch.zeros(2, 3, 32, 32)]))          # float NCHW is not this loop's input


def test_routed_chain_serves_changing_batches_and_extents(E):
    """one RoutedEngine (chain mode) called with batches of different size and extent in a row -- its workspace grows and is reused, the
    weight table is built once -- each result bitwise what per-task engines return"""
    from oracle import params as P
    tasks = {"a": 31, "b": 32, "c": 33}
    routed = E.RoutedEngine("cuda", "bf16")
    single = {}
    for t, seed in tasks.items():
        sd = P.make_state_dict(seed)
        routed.load_task(t, sd)
        single[t] = E.CdanEngine("cuda", "bf16").load(sd)
    names = sorted(tasks)
    for k, (B, H, W) in enumerate([(2, 32, 32), (9, 64, 48), (1, 32, 32), (5, 128, 128), (3, 32, 32)]):
        x = P.lowlight_batch(40 + k, B, H, W)[0].cuda()
        labels = [names[(3 * i + k) % 3] for i in range(B)]
        y = routed.forward(x, labels)
        for i, t in enumerate(labels):
            assert torch.equal(y[i], single[t].forward(x[i:i + 1])[0]), (k, i, t)
    routed.load_task("d", P.make_state_dict(34))          # a new weight set: the table is rebuilt
    single["d"] = E.CdanEngine("cuda", "bf16").load(P.make_state_dict(34))
    x = P.lowlight_batch(50, 4, 32, 32)[0].cuda()
    y = routed.forward(x, ["d", "a", "d", "c"])
    for i, t in enumerate(["d", "a", "d", "c"]):
        assert torch.equal(y[i], single[t].forward(x[i:i + 1])[0]), (i, t)


def test_training_steps_in_flight_match_serial_steps():
    """Two networks training side by side on two streams (different weights, different batches, dropout off so that a step is
    repeatable): loss and all 140 parameter gradients of every overlapped step equal, bit for bit, those of the same step run
    alone -- the training-mode kernels are deterministic and hold up next to another launch on the same CUs (what RCCL's
    all-reduce kernels are to a data-parallel backward)."""
    from models.cdan import CDAN
    from mdie_amd import host as H, synthetic as P
    B, S = 4, 256
    losses = H.build_losses({"enabled": True, "terms": [{"name": "charbonnier", "weight": 1.0}, {"name": "ssim", "weight": 0.5}]})
    nets, xs, ts = [], [], []
    for k in range(2):
        torch.manual_seed(40 + k)
        n = CDAN(precision="bf16").cuda().train()
        n.dropout_p = 0.0
        nets.append(n)
        x, t = P.lowlight_batch(100 + k, B, S, S)
        xs.append(x.cuda())
        ts.append(t.cuda())
    state = [{kk: v.clone() for kk, v in n.state_dict().items()} for n in nets]     # (BatchNorm running statistics move with every forward)

    def step(k):
        nets[k].load_state_dict(state[k])
        nets[k].zero_grad(set_to_none=True)
        total, _ = losses(nets[k](xs[k]), ts[k])
        total.backward()
        return total

    def grads(k):
        return [p.grad.detach().clone() for p in nets[k].parameters() if p.grad is not None]

    ref = []
    for k in range(2):
        l = step(k)
        torch.cuda.synchronize()
        ref.append((l.item(), grads(k)))
    assert len(ref[0][1]) == 140
    streams = [torch.cuda.Stream() for _ in range(2)]
    main = torch.cuda.current_stream()
    for rep in range(6):
        ls = [None, None]
        torch.cuda.synchronize()
        for k in range(2):
            streams[k].wait_stream(main)
            with torch.cuda.stream(streams[k]):
                ls[k] = step(k)
        torch.cuda.synchronize()
        for k in range(2):
            assert ls[k].item() == ref[k][0], (rep, k)
            assert all(torch.equal(a, b) for a, b in zip(grads(k), ref[k][1])), (rep, k)


@pytest.mark.parametrize("prec", ["bf16", "fp16"])
def test_routed_chain_at_size_matches_per_task_engines(E, prec):
    """BASELINE configs[3] at size: 32 images of 256x256, 9 tasks in one launch chain (RoutedEngine), ten times -- every task group's
    output must equal, bit for bit, what a dedicated engine returns for the same group run alone: different weights AND different
    inputs side by side in every launch."""
    from oracle import params as P
    tasks = [f"t{i}" for i in range(9)]
    routed = E.RoutedEngine("cuda", prec)
    sds = {t: P.make_state_dict(20 + i) for i, t in enumerate(tasks)}
    for t in tasks:
        routed.load_task(t, sds[t])
    B = 32
    x, _ = P.lowlight_batch(5, B, 256, 256)
    x = x.cuda()
    labels = [tasks[(i * 7) % 9] for i in range(B)]
    ref = torch.empty_like(x)
    for t in tasks:
        idx = [i for i, l in enumerate(labels) if l == t]
        eng = E.CdanEngine("cuda", prec).load(sds[t])
        ref[idx] = eng.forward(x[idx].contiguous())
        del eng
    torch.cuda.synchronize()
    bad = []
    for rep in range(10):
        y = routed.forward(x, labels)
        torch.cuda.synchronize()
        if not torch.equal(y, ref):
            d = (y - ref).abs().amax(dim=(1, 2, 3))
            bad.append((rep, [(i, labels[i], round(v, 5)) for i, v in enumerate(d.tolist()) if v > 0][:4]))
    assert not bad, f"{len(bad)} of 10 routed forwards differ from the serial groups: {bad[:2]}"


def test_large_image_1024_against_oracle(E):
    """configs[4] (config/pixelation_hard.json 1024x1024 fp16): one 1024x1024 image at its stated dtype, and the fp32 and
    bf16 paths, vs the CPU oracle"""
    from oracle import cdan_oracle as O
    from oracle import params as P
    state_dict = P.make_state_dict(42)
    x, _ = P.lowlight_batch(77, 1, 1024, 1024)
    with torch.no_grad():
        ref = O.cdan_forward(state_dict, x)
    y32 = E.CdanEngine("cuda", "fp32").load(state_dict).forward(x.cuda())
    assert rel_to_max(y32, ref) <= 1e-3                      # north-star tolerance; measured ~1e-5
    y16 = E.CdanEngine("cuda", "bf16").load(state_dict).forward(x.cuda())
    assert rel_to_max(y16, ref) <= BF16_TOL and psnr(y16, ref) >= 46.0        # measured 3.0e-3
    yh = E.CdanEngine("cuda", "fp16").load(state_dict).forward(x.cuda())
    print(f"1024x1024 fp16: rel-to-max {rel_to_max(yh, ref):.3e}, PSNR {psnr(yh, ref):.1f} dB")
    assert rel_to_max(yh, ref) <= F16_OUT_TOL and psnr(yh, ref) >= 70.0      # measured 3.5e-4


@pytest.mark.parametrize("prec", ["bf16", "fp16"])
def test_engines_in_flight_on_different_inputs(E, prec):
    """Three engines alive, two of them running DIFFERENT halves of a batch at the same time on two streams, 40 times: every
    output must equal what the same engine produces alone.  (With the same input on both streams -- bench.py's
    two_batches_in_flight -- a kernel that misbehaves only while another kernel shares its CU goes unseen.  This test found
    one: the splat `v_pk_fma_f32 ... op_sel:[0,1,1]` forms in conv_first_pool_kernel's epilogue, wrong in lanes 48..63 of
    about one tile in 10^4, and only next to a second engine's kernels.)"""
    from mdie_amd import synthetic as P
    sd = P.make_state_dict(42)
    B, S = 32, 256
    x, _ = P.lowlight_batch(1, B, S, S)
    x = x.cuda()
    keep = E.CdanEngine("cuda", prec).load(sd)            # a third engine (and its workspace) alive, as in a serving process
    keep.forward(x)
    n = B // 2
    engs = [E.CdanEngine("cuda", prec).load(sd) for _ in range(2)]
    streams = [torch.cuda.Stream() for _ in range(2)]
    xs = [x[k * n:(k + 1) * n].contiguous() for k in range(2)]
    ref = [engs[k].forward(xs[k]).clone() for k in range(2)]
    torch.cuda.synchronize()
    main = torch.cuda.current_stream()
    bad = []
    for rep in range(40):
        outs = [torch.zeros_like(xs[k]) for k in range(2)]
        torch.cuda.synchronize()
        for k in range(2):
            streams[k].wait_stream(main)
            with torch.cuda.stream(streams[k]):
                engs[k].forward(xs[k], out=outs[k])
        torch.cuda.synchronize()
        bad += [(rep, k, (outs[k] - ref[k]).abs().max().item()) for k in range(2) if not torch.equal(outs[k], ref[k])]
    assert not bad, f"{len(bad)} of 80 overlapped forwards differ from the engine's own serial output: {bad[:4]}"


def test_capture_two_engines_on_forked_streams(E):
    """hipGraph capture of mdie_cdan_forward from streams that are themselves forks inside the capture (two engines, two
    streams), twice in a row with the first capture's engines and graph destroyed in between -- the pattern that took the
    process down in hipStreamEndCapture in round 1, when the library forked its own side streams into the caller's capture.
    Under capture the encoder DenseBlocks are now branches of the graph (capture dependency sets, csrc/engine.hip Branches);
    results must equal the eager ones bit for bit."""
    from oracle import params as P
    sd = P.make_state_dict(42)
    x, _ = P.lowlight_batch(9, 4, 64, 64)
    x = x.cuda()
    dev = x.device
    xs = list(x.chunk(2))
    streams = [torch.cuda.Stream(dev) for _ in range(2)]
    with torch.no_grad():
        ref = E.CdanEngine(dev, "bf16").load(sd).forward(x)
        for round_ in range(2):
            engs = [E.CdanEngine(dev, "bf16").load(sd) for _ in range(2)]     # (the previous round's engines are released here)
            ys = [torch.zeros_like(c) for c in xs]

            def step():
                cur = torch.cuda.current_stream(dev)
                for i in range(2):
                    streams[i].wait_stream(cur)
                    with torch.cuda.stream(streams[i]):
                        engs[i].forward(xs[i], out=ys[i])
                for i in range(2):
                    cur.wait_stream(streams[i])

            step()                                  # eager on the side streams of each engine's aux
            torch.cuda.synchronize()
            assert torch.equal(torch.cat(ys), ref)
            for y in ys:
                y.zero_()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                step()
            torch.cuda.synchronize()
            assert all(float(y.abs().max()) == 0.0 for y in ys)      # capture enqueues nothing
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize()
            assert torch.equal(torch.cat(ys), ref), f"captured replay differs from eager (round {round_})"
            del g, engs


def test_capture_on_origin_stream_has_parallel_branches(E, L):
    """captured from the capture's origin stream (what bench.py does): equal to eager, and the graph really has the three
    DenseBlock branches (more than one root-to-leaf path) unless MDIE_FWD_SERIAL is set"""
    from oracle import params as P
    sd = P.make_state_dict(42)
    x, _ = P.lowlight_batch(10, 2, 64, 64)
    x = x.cuda()
    with torch.no_grad():
        eng = E.CdanEngine(x.device, "fp16").load(sd)
        ref = eng.forward(x)
        y = torch.zeros_like(x)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            eng.forward(x, out=y)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(y, ref)
        eng.use_side_streams = False            # -> MDIE_FWD_SERIAL
        y2 = torch.zeros_like(x)
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2):
            eng.forward(x, out=y2)
        g2.replay()
        torch.cuda.synchronize()
        assert torch.equal(y2, ref)


def test_registered_torch_ops_match_direct_calls(E, net):
    """torch.ops.mdie.cdan_forward / cbam_forward / psnr_ssim == the ctypes path they wrap"""
    import mdie_amd.ops  # noqa: F401
    from mdie_amd import pipeline as PL
    from oracle import params as P
    x, t = P.lowlight_batch(4, 2, 32, 32)
    x, t = x.cuda(), t.cuda()
    eng = net._engine(x.device)
    direct = torch.empty_like(x)
    eng.forward(x, out=direct)                      # explicit-output path: plain ctypes
    via_op = torch.ops.mdie.cdan_forward(x, eng.params, eng._workspace(2, 32, 32), eng.dtype, 0, 0)
    assert torch.equal(via_op, direct) and torch.equal(net(x), direct)
    assert torch.equal(torch.ops.mdie.psnr_ssim(direct, t), PL.psnr_ssim(direct, t))


# ---------------------------------------------------------------------------------------------------------------------
# degradation classifier / router: HIP ResNet18 + heads vs the CPU oracle, seeded random parameters
# ---------------------------------------------------------------------------------------------------------------------
def _router_state_dict(seed=7):
    from mdie_amd import router as R
    from oracle import params as P
    return P.fill_spec(R.router_param_spec(), seed, randomize_bn=True)


@pytest.mark.parametrize("precision,hw,tol", [("fp32", (64, 96), 1e-3), ("fp32", (72, 100), 1e-3), ("bf16", (64, 96), 3e-2), ("fp16", (64, 96), 4e-3)])
def test_router_forward_matches_oracle(E, precision, hw, tol):
    from mdie_amd import router as R
    from oracle import router_oracle as RO
    sd = _router_state_dict()
    g = torch.Generator().manual_seed(hw[0])
    x = torch.rand(3, 3, *hw, generator=g)
    with torch.no_grad():
        rp, rs = RO.classifier_forward({k: v.double() for k, v in sd.items() if v.is_floating_point()}, x.double())
    router = R.DegradationRouter("cuda", precision).load(sd)
    p, s = router.forward(x.cuda())
    assert (p.cpu().double() - rp).abs().max().item() <= tol and (s.cpu().double() - rs).abs().max().item() <= tol


def test_router_checkpoint_format_thresholds_and_routing(E, tmp_path):
    """the reference's checkpoint dict (:860-871) and threshold report (:295-304) drive routing; labels feed RoutedEngine"""
    import json as _json
    from mdie_amd import router as R
    from oracle import params as P
    from oracle import router_oracle as RO
    sd = _router_state_dict(9)
    ckpt = {"model_state": sd, "classes": list(R.CLASSES), "default_thresh": 0.5, "normalize": True,
            "imagenet_mean": list(R.IMAGENET_MEAN), "imagenet_std": list(R.IMAGENET_STD), "epoch": 3}
    router = R.DegradationRouter("cuda", "fp32").load(ckpt)
    x, _ = P.lowlight_batch(21, 6, 64, 64)
    probs, _ = router.forward(x.cuda())
    # thresholds chosen from the observed probabilities so that some images route and some pass through
    pc = probs.cpu()
    thr = {c: float(pc[:, i].median()) + 0.02 for i, c in enumerate(R.CLASSES)}
    path = os.path.join(str(tmp_path), "thresholds_val.json")
    with open(path, "w") as fh:
        _json.dump({"objective": "test", "thresholds": thr}, fh)
    router.load_thresholds(path)
    labels, _ = router.route(x.cuda())
    with torch.no_grad():
        rp, _ = RO.classifier_forward(sd, x)
    assert labels == RO.route(rp, [thr[c] for c in R.CLASSES], list(R.CLASSES))
    # end to end: routed enhancement with pass-through for undetected images
    routed = E.RoutedEngine("cuda", "bf16")
    for i, c in enumerate(R.CLASSES):
        if c in labels:
            routed.load_task(c, P.make_state_dict(200 + i))
    y = routed.forward(x.cuda(), labels)
    for i, t in enumerate(labels):
        if t is None:
            assert torch.equal(y[i], x[i].cuda())
        else:
            assert not torch.equal(y[i], x[i].cuda())
    with pytest.raises(Exception):
        R.DegradationRouter("cuda").load({k: v for k, v in sd.items() if "layer3" not in k})
    with pytest.raises(Exception):
        router.forward(x)                        # CPU tensor


_ORACLE_TRAIN_STEP = {}


@pytest.mark.parametrize("precision,shape,min_cos,med_cos,out_tol", [("fp32", (2, 64, 64), 0.9999, 0.99999, 2e-4),
                                                                     ("fp32", (1, 40, 56), 0.9999, 0.99999, 2e-4),    # one image, ragged tiles
                                                                     # 1x1 maps at the deep end: BatchNorm over TWO samples normalises to exactly +-1, the true gradient
                                                                     # through it is ~0 and what is left is summation-order noise -> direction only loosely pinned there
                                                                     ("fp32", (2, 8, 8), 0.95, 0.9999, 2e-4),
                                                                     ("bf16", (2, 64, 64), 0.87, 0.98, 4e-2),     # measured: worst 0.8934-0.9176, median 0.9870, output 2.9e-2 .. 3.1e-2 (moves with the fp32 summation ORDER of the CBAM gate's hidden layer: 128-sample BatchNorms downstream)
                                                                     # fp16 = the reference's own autocast dtype: gradients need its GradScaler (models/model.py:31,164)
                                                                     ("fp16", (2, 64, 64), 0.986, 0.9975, 8e-3),      # measured: worst 0.9931, median 0.99874, output 3.9e-3
                                                                     # 256x256: the kernel selection of BASELINE configs[2] (512x512, B=8/GPU) -- B=2: encoder.conv2 has 2*8*4*2 = 128
                                                                     # items -> conv_wide_kernel<ACT_NONE> (forward and dgrad of the wide layers), 512 full-resolution tiles -> conv_kernel;
                                                                     # B=4: 1024 tiles -> conv_thin_kernel for the final DenseBlock's layers and their input gradients
                                                                     # Output bounds at this size: batch-statistic BatchNorm amplifies storage rounding much more than the eval-mode
                                                                     # network does.  torch's OWN mixed precision on this network (the oracle under torch.autocast("cpu"), training
                                                                     # mode, against the same fp64 run; round 3, this container) is off by 9.9e-2 (bf16) / 7.8e-3 (fp16) of max at
                                                                     # 2x256x256 and 1.0e-1 / 1.4e-2 at 4x256x256; the engine measured 7.3e-2 / <8e-3 and 1.9e-1 / 1.05e-2.
                                                                     # (bf16's gradient quality depends on the batch: measured median 0.940 / worst 0.746 here, 0.989 / 0.905 at B = 4 on
                                                                     #  the SAME kernels -- fp16 on this very shape holds 0.99985 / 0.9955, so the kernels the shape selects are right and what
                                                                     #  is left is bf16's 8-bit mantissa under batch-statistic BatchNorm: README recommends fp16 + GradScaler for training)
                                                                     ("bf16", (2, 256, 256), 0.70, 0.92, 1.2e-1),
                                                                     ("fp16", (2, 256, 256), 0.98, 0.996, 9e-3),     # output: 7.9e-3 .. 8.05e-3 with the glue kernels' fp32 math packed / unpacked (torch's own autocast: 7.8e-3)
                                                                     ("bf16", (4, 256, 256), 0.85, 0.975, 2.5e-1),
                                                                     ("fp16", (4, 256, 256), 0.95, 0.99, 1.6e-2)])      # measured: worst 0.9601 (encoder.dense2.layers.1.0.bias), median 0.9922
def test_whole_network_training_step_vs_oracle(E, precision, shape, min_cos, med_cos, out_tol):
    """forward + backward of the whole network in training mode (batch-stat BN, dropout off) at 2x3x64x64 against the CPU
    oracle differentiated by autograd: output, loss, and the direction of EVERY parameter gradient (cosine similarity;
    bf16 stores activations AND gradient tensors in bf16 -- like autocast training -- and at this size the deep layers
    normalise over 128 samples, so its gradients are compared by direction: measured median 0.98, worst 0.89 at the
    bottleneck's gate MLP, whose arg-max / ReLU routing can flip under bf16 rounding)."""
    from models.cdan import CDAN
    from oracle import cdan_oracle as O
    from oracle import params as P
    scalar_tol = {"fp32": 1e-3, "fp16": 0.03, "bf16": 0.15}[precision]      # measured: <= 6e-4 / 0.011 / 0.06
    sd = P.make_state_dict(42)
    x, t = P.lowlight_batch(77, *shape)
    if shape not in _ORACLE_TRAIN_STEP:          # fp64 oracle, once per shape (the 256x256 cases share it between precisions)
        ref_sd = {k: (v.clone().double().requires_grad_(True) if v.is_floating_point() and "running" not in k else (v.double() if v.is_floating_point() else v))
                  for k, v in sd.items()}
        ry = O.cdan_forward(ref_sd, x.double(), "train", {})
        rloss = torch.sqrt((ry - t.double()) ** 2 + 1e-6).mean()
        rloss.backward()
        _ORACLE_TRAIN_STEP.clear()               # keep one shape at a time (gradients of the whole network in fp64)
        _ORACLE_TRAIN_STEP[shape] = ({k: v.grad for k, v in ref_sd.items() if getattr(v, "grad", None) is not None}, ry.detach(), rloss.detach())
    ref_grad, ry, rloss = _ORACLE_TRAIN_STEP[shape]
    net = CDAN(precision=precision)
    net.load_state_dict(sd, strict=True)
    net = net.cuda().train()
    net.dropout_p = 0.0
    y = net(x.cuda())
    loss = torch.sqrt((y - t.cuda()) ** 2 + 1e-6).mean()
    scale = 65536.0 if precision == "fp16" else 1.0      # torch.cuda.amp.GradScaler's initial scale: fp16 gradient TENSORS would underflow without it
    (loss * scale).backward()
    for p in net.parameters():
        p.grad /= scale
    out_err = rel_to_max(y, ry)
    worst, allcos, scalars = (1.0, None), [], []
    for k, p in net.named_parameters():
        g, r = p.grad.detach().double().cpu().reshape(-1), ref_grad[k].reshape(-1)
        if r.norm().item() < 1e-9 * max(1.0, float(r.numel()) ** 0.5):   # biases in front of a batch-stat BatchNorm: exact zeros
            assert g.abs().max().item() <= 1e-6, k
            continue
        if g.numel() == 1:      # a scalar has no direction: the four CBAM spatial-gate BatchNorm(1) weights / biases are held by value below
            scalars.append((k, g.item(), r.item()))
            continue
        cos = float(torch.dot(g, r) / (g.norm() * r.norm()).clamp_min(1e-30))
        allcos.append(cos)
        if cos < worst[0]:
            worst = (cos, k)
    median = sorted(allcos)[len(allcos) // 2]
    print(f"[{precision} {shape}] output {out_err:.3e} of max, loss {loss.item():.6f} vs {rloss.item():.6f}; gradient cosine: median {median:.5f}, worst {worst[0]:.5f} at {worst[1]}")
    print("    scalar gradients (engine / oracle): " + ", ".join(f"{k.replace('.SpatialGate.spatial.bn', '.sbn')} {a:+.3e} / {b:+.3e}" for k, a, b in scalars))
    assert out_err <= out_tol
    assert loss.item() == pytest.approx(rloss.item(), rel=out_tol)
    assert worst[0] >= min_cos and median >= med_cos, (worst, median)
    assert len(scalars) == 8
    sc_scale = max(abs(b) for _, _, b in scalars)
    for k, a, b in scalars:          # within scalar_tol of the largest of them (a BatchNorm(1) bias gradient is a sum over a whole map that may cancel to ~0)
        assert abs(a - b) <= scalar_tol * sc_scale, (k, a, b)


def test_captured_training_step_equals_eager_steps(E):
    """train.CapturedStep: forward + loss + backward + Adam as ONE hipGraph.  Two nets from the same checkpoint and the same
    dropout salt take 3 steps on changing batches, one eagerly, one by graph replay: parameters, BatchNorm buffers and
    losses must agree bit for bit (the dropout counter lives on the device, so every replay draws the masks the eager step
    of the same index draws), and building the graph must not have moved the training state."""
    from models.cdan import CDAN
    from mdie_amd import host as H
    from mdie_amd import train as T
    from oracle import params as P
    sd = P.make_state_dict(42)
    losses = H.build_losses({"enabled": True, "terms": [{"name": "charbonnier", "weight": 1.0}, {"name": "ssim", "weight": 0.5}]})
    batches = [tuple(v.cuda() for v in P.lowlight_batch(60 + i, 2, 32, 32)) for i in range(3)]
    runs = []
    for mode in ("eager", "graph"):
        torch.manual_seed(5)
        net = CDAN(precision="bf16")
        net.load_state_dict(sd, strict=True)
        net = net.cuda().train()
        opt = torch.optim.Adam(net.parameters(), lr=1e-3, capturable=True)
        vals = []
        if mode == "graph":
            before = [b.clone() for b in net.buffers()] + [p.detach().clone() for p in net.parameters()]
            step = T.CapturedStep(net, losses, opt, *batches[0])
            after = [b for b in net.buffers()] + [p.detach() for p in net.parameters()]
            assert all(torch.equal(a, b) for a, b in zip(before, after)), "building the graph moved the training state"
            for x, t in batches:
                vals.append(step(x, t).clone())
        else:
            for x, t in batches:
                opt.zero_grad(set_to_none=True)
                total, v = losses(net(x), t)
                total.backward()
                opt.step()
                vals.append(v.clone())
        torch.cuda.synchronize()
        runs.append((vals, [p.detach().clone() for p in net.parameters()], [b.clone() for b in net.buffers()]))
    assert all(torch.equal(a, b) for a, b in zip(runs[0][0], runs[1][0])), "loss values differ"
    assert all(torch.equal(a, b) for a, b in zip(runs[0][1], runs[1][1])), "parameters differ after 3 steps"
    assert all(torch.equal(a, b) for a, b in zip(runs[0][2], runs[1][2])), "BatchNorm buffers differ"
    assert not torch.equal(runs[1][0][0], runs[1][0][1])


@pytest.mark.parametrize("whole", [True, False])
def test_captured_steps_two_batch_shapes_and_eager_mix(E, whole):
    """What `Model.train_step` does with a dataset whose size is not a multiple of the batch size (drop_last=False,
    models/model.py:154-172 over data/dataset.py's loader): one CapturedStep per batch shape, the second one built in the
    MIDDLE of training (Adam's moments populated), replays of the two alternating, an eager step in between.  `whole`:
    Adam inside the graph (single GPU, no GradScaler); otherwise the graph holds forward + loss + backward and the optimizer
    steps on `p.grad` after the replay (the GradScaler / gradient-exchange path) -- that path reads `p.grad`, which must be
    the gradient of the replay just made, not of whichever graph was captured last.  Parameters, buffers, losses and the
    evaluation output afterwards bit-identical to the same schedule stepped eagerly."""
    from models.cdan import CDAN
    import mdie_amd.host as H
    import mdie_amd.train as T
    from oracle import params as P
    sd = P.make_state_dict(42)
    losses = H.build_losses({"enabled": True, "terms": [{"name": "charbonnier", "weight": 1.0}]})
    full = [tuple(v.cuda() for v in P.lowlight_batch(80 + i, 3, 32, 32)) for i in range(5)]
    part = [tuple(v.cuda() for v in P.lowlight_batch(90 + i, 2, 32, 32)) for i in range(2)]
    schedule = [("g", full[0]), ("g", full[1]), ("g", part[0]), ("g", full[2]), ("e", full[3]), ("g", part[1]), ("g", full[4])]
    xe = P.lowlight_batch(99, 2, 32, 32)[0].cuda()
    runs = []
    for mode in ("eager", "graph"):
        torch.manual_seed(5)
        net = CDAN(precision="bf16")
        net.load_state_dict(sd, strict=True)
        net = net.cuda().train()
        opt = torch.optim.Adam(net.parameters(), lr=1e-3, capturable=whole)
        captured, vals = {}, []
        for kind, (x, t) in schedule:
            if mode == "graph" and kind == "g":
                key = tuple(x.shape)
                if key not in captured:
                    captured[key] = T.CapturedStep(net, losses, opt if whole else None, x, t)
                vals.append(captured[key](x, t).clone())
                if not whole:
                    opt.step()
            else:
                opt.zero_grad(set_to_none=True)
                total, v = losses(net(x), t)
                total.backward()
                opt.step()
                vals.append(v.clone())
        net.eval()
        with torch.no_grad():
            ye = net(xe).clone()               # must run on the weights the replays produced (no tensor version was bumped by them)
        torch.cuda.synchronize()
        runs.append((vals, [p.detach().clone() for p in net.parameters()], [b.clone() for b in net.buffers()], ye))
        if mode == "graph":
            assert len(captured) == 2
    for i, (a, b) in enumerate(zip(runs[0][0], runs[1][0])):
        assert torch.equal(a, b), f"loss values differ at step {i}"
    assert all(torch.equal(a, b) for a, b in zip(runs[0][1], runs[1][1])), "parameters differ"
    assert all(torch.equal(a, b) for a, b in zip(runs[0][2], runs[1][2])), "BatchNorm buffers differ"
    assert torch.equal(runs[0][3], runs[1][3]), "evaluation after graph-replayed training ran on stale packed weights"


def test_training_step_is_bitwise_reproducible(E):
    """same seed, same batch -> bit-identical loss and gradients (ordered reductions everywhere, counter-based dropout)"""
    from models.cdan import CDAN
    from oracle import params as P
    sd = P.make_state_dict(42)
    x, t = P.lowlight_batch(5, 4, 64, 64)
    x, t = x.cuda(), t.cuda()
    runs = []
    for _ in range(2):
        torch.manual_seed(123)
        net = CDAN(precision="bf16")
        net.load_state_dict(sd, strict=True)
        net = net.cuda().train()
        loss = torch.sqrt((net(x) - t) ** 2 + 1e-6).mean()
        loss.backward()
        runs.append((loss.detach().clone(), [p.grad.clone() for p in net.parameters()], [b.clone() for b in net.buffers()]))
    assert torch.equal(runs[0][0], runs[1][0])
    assert all(torch.equal(a, b) for a, b in zip(runs[0][1], runs[1][1]))
    assert all(torch.equal(a, b) for a, b in zip(runs[0][2], runs[1][2]))


@pytest.mark.parametrize("shape", [(1, 8, 8), (3, 8, 16), (2, 16, 8), (5, 24, 40), (1, 72, 8), (2, 8, 136)])
def test_small_and_ragged_extents_against_oracle(E, net, shape):
    """smallest legal extents (the bottleneck is 1x1 at 8x8 input), odd batch sizes, strongly non-square pictures"""
    from oracle import cdan_oracle as O
    from oracle import params as P
    x, _ = P.lowlight_batch(sum(shape), *shape)
    with torch.no_grad():
        ref = O.cdan_forward(P.make_state_dict(42), x)
        net.precision = "fp32"
        y = net(x.cuda())
        assert rel_to_max(y, ref) <= CONTRACT_TOL
        net.precision = "bf16"
        yb = net(x.cuda())
        net.precision = "fp16"
        yh = net(x.cuda())
    assert rel_to_max(yb, ref) <= BF16_TOL          # measured <= 2.4e-3
    assert rel_to_max(yh, ref) <= F16_OUT_TOL


# ---- round 3: the entry points the faster training step added ---------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("C_new,off,C_fold,c_real,split,gap", [(64, 0, 64, 64, 64, 0), (16, 32, 48, 35, 3, 13), (16, 0, 16, 3, 3, 13)])
def test_bn_stats_fold_equals_stats_then_fold(E, L, precision, C_new, off, C_fold, c_real, split, gap):
    """mdie_bn_stats_fold (statistics pass + ONE final/fold launch) against mdie_bn_stats followed by mdie_bn_fold: bit-identical
    mean / var / scale / shift / invstd / running statistics, for a plain BatchNorm, for a DenseBlock layer whose new map joins
    earlier ones (statistics already in place), and with the final_dense channel gap."""
    import ctypes as C
    dt = E.dtype_id(precision)
    g = torch.Generator().manual_seed(11)
    x = (torch.randn(2, C_new, 12, 20, generator=g) * 1.7 + 0.3).cuda().to(TORCH_DT[precision]).contiguous(memory_format=torch.channels_last)
    N = 2 * 12 * 20
    gamma, beta = torch.randn(c_real, generator=g).cuda(), torch.randn(c_real, generator=g).cuda()
    old = torch.rand(2, C_fold, generator=g).cuda() + 0.1                 # statistics of the earlier maps (mean, var)
    sp = torch.cuda.current_stream().cuda_stream

    def run(fused):
        mv = old.clone()
        rm, rv = torch.zeros(c_real, device="cuda"), torch.ones(c_real, device="cuda")
        k = torch.full((3, C_fold), 7.0, device="cuda")
        nws = L.lib.mdie_bn_workspace_bytes(C_new)
        ws = torch.empty(nws, dtype=torch.uint8, device="cuda")
        if fused:
            d = L.BnStatsFoldDesc()
            d.dtype, d.N, d.x, d.C, d.stride = dt, N, x.data_ptr(), C_new, C_new
            d.mean, d.var = mv[0, off:].data_ptr(), mv[1, off:].data_ptr()
            d.workspace, d.workspace_bytes = ws.data_ptr(), nws
            d.C_fold, d.C_real, d.split, d.gap = C_fold, c_real, split, gap
            d.fold_mean, d.fold_var, d.gamma, d.beta, d.eps, d.momentum = mv[0].data_ptr(), mv[1].data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-5, 0.1
            d.running_mean, d.running_var = rm.data_ptr(), rv.data_ptr()
            d.scale, d.shift, d.invstd = k[0].data_ptr(), k[1].data_ptr(), k[2].data_ptr()
            L.check(L.lib.mdie_bn_stats_fold(C.byref(d), sp), "mdie_bn_stats_fold")
        else:
            L.check(L.lib.mdie_bn_stats(dt, N, x.data_ptr(), C_new, C_new, mv[0, off:].data_ptr(), mv[1, off:].data_ptr(), ws.data_ptr(), nws, sp), "mdie_bn_stats")
            L.check(L.lib.mdie_bn_fold(C_fold, c_real, split, gap, mv[0].data_ptr(), mv[1].data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-5, 0.1, N,
                                       rm.data_ptr(), rv.data_ptr(), k[0].data_ptr(), k[1].data_ptr(), k[2].data_ptr(), sp), "mdie_bn_fold")
        torch.cuda.synchronize()
        return mv, rm, rv, k

    for a, b in zip(run(True), run(False)):
        assert torch.equal(a, b)
    mv = run(True)[0]
    ref = x.float().permute(1, 0, 2, 3).reshape(C_new, -1)
    assert rel_to_max(mv[0, off:off + C_new], ref.mean(1)) <= 1e-5 and rel_to_max(mv[1, off:off + C_new], ref.var(1, unbiased=False)) <= 1e-4


@pytest.mark.gpu
def test_bn_stats_fold_rejects_bad_arguments(L):
    import ctypes as C
    d = L.BnStatsFoldDesc()
    assert L.lib.mdie_bn_stats_fold(None, None) == -1
    x = torch.zeros(1, 16, 8, 8, device="cuda")
    mv, k = torch.zeros(2, 32, device="cuda"), torch.zeros(3, 32, device="cuda")
    ws = torch.empty(L.lib.mdie_bn_workspace_bytes(16), dtype=torch.uint8, device="cuda")
    d.dtype, d.N, d.x, d.C, d.stride = L.F32, 64, x.data_ptr(), 16, 16
    d.mean, d.var, d.workspace, d.workspace_bytes = mv[0, 24:].data_ptr(), mv[1, 24:].data_ptr(), ws.data_ptr(), ws.numel()
    d.C_fold, d.C_real, d.split, d.gap = 32, 32, 32, 0
    d.fold_mean, d.fold_var, d.gamma, d.beta = mv[0].data_ptr(), mv[1].data_ptr(), k[0].data_ptr(), k[1].data_ptr()
    d.scale, d.shift, d.invstd = k[0].data_ptr(), k[1].data_ptr(), k[2].data_ptr()
    assert L.lib.mdie_bn_stats_fold(C.byref(d), None) == -1 and b"inside the fold range" in L.lib.mdie_last_error()   # 24 + 16 > 32
    d.mean, d.var, d.workspace_bytes = mv[0, 16:].data_ptr(), mv[1, 16:].data_ptr(), 16
    assert L.lib.mdie_bn_stats_fold(C.byref(d), None) == -3


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("ks", [3, 1])
def test_pack_job_places_real_channels_inside_stored_ones(E, L, precision, ks):
    """mdie_pack_conv_weight_job with an input-channel gap (forward form) and an output-channel gap (input-gradient form) against the
    pack of the explicitly zero-padded weight (what train.py built with torch.cat every step before): byte-identical; and the batch
    entry over both jobs gives the same bytes."""
    import ctypes as C
    dt = E.dtype_id(precision)
    g = torch.Generator().manual_seed(5)
    cout, real_c, gap, extra = 16, 3, 13, 32
    cin_real, cin_st = real_c + extra, real_c + gap + extra
    w = torch.randn(cout, cin_real, ks, ks, generator=g).cuda()
    wpad = torch.cat((w[:, :real_c], w.new_zeros(cout, gap, ks, ks), w[:, real_c:]), 1).contiguous()
    sp = torch.cuda.current_stream().cuda_stream

    def job(src, transposed, co, ci, co_st, ci_st, split, gp, osplit, ogap):
        dst = torch.zeros(L.lib.mdie_conv_weight_bytes(dt, ks, ci_st, co_st), dtype=torch.uint8, device="cuda")
        return L.PackJob(src.data_ptr(), dst.data_ptr(), ks, transposed, co, ci, co_st, ci_st, split, gp, osplit, ogap), dst

    def run(j):
        L.check(L.lib.mdie_pack_conv_weight_job(dt, C.byref(j[0]), sp), "mdie_pack_conv_weight_job")
        torch.cuda.synchronize()
        return j[1]

    fwd = run(job(w, 0, cout, cin_real, cout, cin_st, real_c, gap, cout, 0))
    fwd_ref = run(job(wpad, 0, cout, cin_st, cout, cin_st, cin_st, 0, cout, 0))
    assert torch.equal(fwd, fwd_ref) and fwd.any()
    # input-gradient form: the dgrad convolution's OUTPUT channels are the layer's stored input channels
    dg = run(job(w, 1, cin_real, cout, cin_st, cout, cout, 0, real_c, gap))
    dg_ref = run(job(wpad, 1, cin_st, cout, cin_st, cout, cout, 0, cin_st, 0))
    assert torch.equal(dg, dg_ref) and dg.any()
    j1, j2 = job(w, 0, cout, cin_real, cout, cin_st, real_c, gap, cout, 0), job(w, 1, cin_real, cout, cin_st, cout, cout, 0, real_c, gap)
    table = torch.frombuffer(bytearray(bytes(j1[0]) + bytes(j2[0])), dtype=torch.uint8).cuda()
    L.check(L.lib.mdie_pack_conv_weights_batch(dt, table.data_ptr(), 2, sp), "mdie_pack_conv_weights_batch")
    torch.cuda.synchronize()
    assert torch.equal(j1[1], fwd_ref) and torch.equal(j2[1], dg_ref)
    bad = job(w, 0, cout, cin_real, cout, cin_st - 16, real_c, gap, cout, 0)[0]
    assert L.lib.mdie_pack_conv_weight_job(dt, C.byref(bad), sp) == -1


@pytest.mark.gpu
def test_zero_bias_gradients_are_separate_slices_of_one_arena(E):
    """train._zero_grad_vec: the 24 exactly-zero bias gradients of a step are slices of one zero-filled tensor -- distinct memory per
    parameter (an in-place user of .grad such as GradScaler.unscale_ or gradient clipping must not touch a neighbour), all zero, and a
    request past the arena's end falls back to its own tensor."""
    import mdie_amd.train as T
    net, _ = _train_net()
    net = net.cuda().train()
    x = torch.rand(2, 3, 32, 32, device="cuda")
    net.zero_grad(set_to_none=True)
    net(x).mean().backward()
    zero = [(n, p) for n, p in net.named_parameters() if n.endswith("bias") and p.grad is not None and p.dim() == 1 and not p.grad.any()]
    assert len(zero) == 24          # 4 encoder convolutions + 4 x 4 dense layers + 4 decoder stages: every bias that meets a BatchNorm
    spans = sorted((p.grad.data_ptr(), p.grad.data_ptr() + 4 * p.grad.numel()) for _, p in zero)
    assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:]))
    zero[0][1].grad.add_(1.0)
    assert all(not p.grad.any() for _, p in zero[1:])
    T._new_zero_arena(x.device, n=8)
    a, b = T._zero_grad_vec(8, x.device), T._zero_grad_vec(8, x.device)
    assert a.data_ptr() != b.data_ptr() and not a.any() and not b.any() and b.numel() == 8


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("pool,p", [(True, 0.2), (True, 0.0), (False, 0.2)])
def test_pool_bwd_two_pass_equals_masked_gradient_then_apply(E, L, precision, pool, p):
    """mdie_bn_act_pool_bwd with two_pass = 1 (sums-only pass, fold, a pass that stores dL/dy directly) against the original form
    (masked gradient dz stored, mdie_bn_bwd_apply over it): same dgamma / dbeta / coef bit for bit (the sums never saw the stored
    dz), and dL/dy equal up to the one rounding of dz to the storage type that the two-pass form no longer makes."""
    import ctypes as C
    dt = E.dtype_id(precision)
    td = TORCH_DT[precision]
    g = torch.Generator().manual_seed(21)
    B, H, W, Cc = 2, 12, 20, 64
    cl = lambda t: t.cuda().to(td).contiguous(memory_format=torch.channels_last)
    y = cl(torch.randn(B, Cc, H, W, generator=g))
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    d_o, d_t = cl(torch.randn(B, Cc, Ho, Wo, generator=g)), cl(torch.randn(B, Cc, Ho, Wo, generator=g))
    mean, var = y.float().mean((0, 2, 3)), y.float().var((0, 2, 3), unbiased=False)
    gamma, beta = torch.rand(Cc, generator=g).cuda() + 0.5, torch.randn(Cc, generator=g).cuda() * 0.1
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    scale = gamma * invstd
    shift = beta - mean * scale
    sp = torch.cuda.current_stream().cuda_stream

    def run(two_pass):
        dz = torch.full_like(y, 3.0)
        dgb, coef = torch.zeros(2, Cc, device="cuda"), torch.zeros(2, Cc, device="cuda")
        ws = torch.empty(L.lib.mdie_bn_workspace_bytes(Cc), dtype=torch.uint8, device="cuda")
        d = L.BnPoolBwdDesc()
        d.dtype, d.B, d.H, d.W, d.C, d.c_real = dt, B, H, W, Cc, Cc
        d.y, d.y_stride = y.data_ptr(), Cc
        d.scale, d.shift, d.mean, d.invstd, d.pool = scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr(), int(pool)
        d.d_out, d.d_out_stride, d.d_drop, d.d_drop_stride = d_o.data_ptr(), Cc, d_t.data_ptr(), Cc
        d.p, d.seed, d.seed_dev = p, 1234, None
        d.dz, d.dz_stride = dz.data_ptr(), Cc
        d.dgamma, d.dbeta, d.coef = dgb[0].data_ptr(), dgb[1].data_ptr(), coef.data_ptr()
        d.workspace, d.workspace_bytes, d.two_pass = ws.data_ptr(), ws.numel(), two_pass
        L.check(L.lib.mdie_bn_act_pool_bwd(C.byref(d), sp), "mdie_bn_act_pool_bwd")
        if not two_pass:
            a = L.BnBwdDesc()
            a.dtype, a.N, a.nseg = dt, B * H * W, 1
            a.x[0], a.g[0] = L.Seg(y.data_ptr(), Cc, Cc), L.Seg(dz.data_ptr(), Cc, Cc)
            a.accumulate, a.da, a.da_stride = 0, dz.data_ptr(), Cc
            a.mean, a.invstd, a.scale, a.shift, a.relu, a.coef = mean.data_ptr(), invstd.data_ptr(), scale.data_ptr(), shift.data_ptr(), 0, coef.data_ptr()
            L.check(L.lib.mdie_bn_bwd_apply(C.byref(a), sp), "mdie_bn_bwd_apply")
        torch.cuda.synchronize()
        return dz.float(), dgb, coef

    dz2, dgb2, coef2 = run(1)
    dz0, dgb0, coef0 = run(0)
    assert torch.equal(dgb2, dgb0) and torch.equal(coef2, coef0)
    assert rel_to_max(dz2, dz0) <= {"fp32": 1e-6, "fp16": 2e-3, "bf16": 1.6e-2}[precision]
    assert float(dz2.abs().max()) > 0.1


@pytest.mark.gpu
@pytest.mark.parametrize("precision,shape", [("bf16", (4, 64, 64)), ("fp16", (2, 128, 128)), ("bf16", (2, 256, 256)), ("bf16", (8, 512, 512))])
def test_side_stream_weight_gradients_equal_main_stream(E, precision, shape, monkeypatch):
    """train.WGRAD_STREAM: weight gradients launched on a stream of their own (forked at the call, joined once when backward ends)
    against the same three optimizer steps with everything on one stream: losses, parameters and buffers bit-identical -- any
    difference is a race (a dW read before its kernel finished, an input overwritten or recycled while the side stream read it)."""
    import mdie_amd.train as T
    from models.cdan import CDAN
    from oracle import params as P
    sd = P.make_state_dict(42)
    batches = [tuple(v.cuda() for v in P.lowlight_batch(5 + i, *shape)) for i in range(3)]

    monkeypatch.setattr(T, "WGRAD_STREAM_MIN_PIXELS", 0)       # (the small shapes too: by default only steps >= 8 x 384 x 384 pixels fork)

    def run(side):
        monkeypatch.setattr(T, "WGRAD_STREAM", side)
        torch.manual_seed(123)
        net = CDAN(precision=precision)
        net.load_state_dict(sd, strict=True)
        net = net.cuda().train()
        opt = torch.optim.Adam(net.parameters(), lr=1e-3, fused=True)
        losses = []
        for x, t in batches:
            opt.zero_grad(set_to_none=True)
            loss = torch.sqrt((net(x) - t) ** 2 + 1e-6).mean()
            loss.backward()
            opt.step()
            losses.append(loss.detach().clone())
        torch.cuda.synchronize()
        return losses, [p.detach().clone() for p in net.parameters()], [b.clone() for b in net.buffers()]

    b = run(False)
    names = [n for n, _ in CDAN().named_parameters()]
    a = run(True)      # ONCE: a test is not a hunt (the opt-in schedule's open finding: profiles/LEDGER.md (rounds 1-4) section 4, finding 6)
    assert all(torch.equal(u, v) for u, v in zip(a[0], b[0]))
    bad = [n for n, u, v in zip(names, a[1], b[1]) if not torch.equal(u, v)]
    assert not bad, f"parameters differ: {bad[:6]}"
    assert all(torch.equal(u, v) for u, v in zip(a[2], b[2]))


@pytest.mark.gpu
@pytest.mark.parametrize("form", [2, 1])
def test_training_step_is_bit_identical_in_the_cu_sharing_forms_of_its_wide_layers(E, monkeypatch, form):
    """MDIE_TRAIN_SHARE_CU (train.WIDE_SHARE_CU): the training step's conv_wide layers in two shorter runs per CU (2) or on conv_kernel (1) instead of
    one persistent workgroup per CU -- the switch a multi-GPU run A/Bs against exposed all-reduce time.  A schedule, not arithmetic: loss, every
    gradient and the updated BatchNorm statistics of one step at a shape conv_wide takes (8 x 256 x 256) must not change by a bit."""
    import mdie_amd.train as T
    from models.cdan import CDAN
    from oracle import params as P
    sd = P.make_state_dict(42)
    x, t = (v.cuda() for v in P.lowlight_batch(61, 8, 256, 256))

    def run(share):
        monkeypatch.setattr(T, "WIDE_SHARE_CU", share)
        torch.manual_seed(7)
        net = CDAN(precision="bf16")
        net.load_state_dict(sd, strict=True)
        net = net.cuda().train()
        loss = torch.sqrt((net(x) - t) ** 2 + 1e-6).mean()
        loss.backward()
        torch.cuda.synchronize()
        return loss.detach().clone(), [p.grad.clone() for p in net.parameters()], [b.clone() for b in net.buffers()]

    a, b = run(0), run(form)
    assert torch.equal(a[0], b[0])
    bad = [i for i, (u, v) in enumerate(zip(a[1], b[1])) if not torch.equal(u, v)]
    assert not bad, f"{len(bad)} gradients differ between the forms"
    assert all(torch.equal(u, v) for u, v in zip(a[2], b[2]))


@pytest.mark.gpu
def test_ddp_one_rank_nccl_gradients_live_in_the_buckets(E, monkeypatch):
    """The data-parallel step on RCCL (a ONE-rank "nccl" group: the one-GPU box has no second device, the code path -- grad hooks ->
    bucket complete -> asynchronous all-reduce on the communication stream -> finish -- is the one eight ranks run; the step being
    sharded: models/model.py:154-166).  With GradBuckets active every parameter gradient of the network is written by its kernel
    into its bucket slice: `.grad` is a view of the flat bucket before and after the exchange, nothing is copied, and losses,
    gradients and parameters of two optimizer steps are bit-identical to the same steps without the exchange (world size 1:
    the average of one).  Also with the weight gradients on their side stream, and for a replayed CapturedStep + exchange()."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "ddp_one_rank_gpu.py")], capture_output=True, text=True, timeout=600)
    ok = "DDP-ONE-RANK-OK" in r.stdout and "DDP-TEARDOWN-OK" in r.stdout and r.returncode == 0
    if not ok:
        # the child's WHOLE output goes to a file (round 5 kept 3 000 characters of a C++ exception's stack and lost its text)
        out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "gpurun_out", "ddp_one_rank_child.log")
        try:
            os.makedirs(os.path.dirname(out), exist_ok=True)
            with open(out, "w") as f:
                f.write(f"rc {r.returncode}\n--- stdout\n{r.stdout}\n--- stderr\n{r.stderr}\n")
        except OSError:
            out = "(could not be written)"
        what = "did not finish its checks" if "DDP-ONE-RANK-OK" not in r.stdout else "passed its checks but its process-group teardown did not end cleanly"
        pytest.fail("tests/ddp_one_rank_gpu.py %s (rc %d; full output: %s)\n--- stdout\n%s\n--- stderr\n%s" % (what, r.returncode, out, r.stdout[-3000:], r.stderr[-8000:]), pytrace=False)


@pytest.mark.gpu
def test_side_stream_second_gradient_of_a_parameter_stays_on_the_main_stream(E, monkeypatch):
    """The network applied TWICE before one backward: every parameter receives two gradients, which autograd sums on the main
    stream.  Only the first dW of a backward may run on the side stream; the second waits for it and runs on the main stream
    (train._first_sighting) -- gradients bit-identical to the single-stream schedule."""
    import mdie_amd.train as T
    from models.cdan import CDAN
    from oracle import params as P
    sd = P.make_state_dict(42)
    (x1, t1), (x2, t2) = [tuple(v.cuda() for v in P.lowlight_batch(11 + i, 2, 64, 64)) for i in range(2)]
    monkeypatch.setattr(T, "WGRAD_STREAM_MIN_PIXELS", 0)

    def run(side):
        monkeypatch.setattr(T, "WGRAD_STREAM", side)
        net = CDAN(precision="bf16")
        net.load_state_dict(sd, strict=True)
        net = net.cuda().train()
        net.dropout_p = 0.0
        loss = torch.sqrt((net(x1) - t1) ** 2 + 1e-6).mean() + torch.sqrt((net(x2) - t2) ** 2 + 1e-6).mean()
        loss.backward()
        torch.cuda.synchronize()
        return [p.grad.clone() for p in net.parameters()]

    a, b = run(True), run(False)
    assert all(g is not None for g in a)
    bad = [i for i, (u, v) in enumerate(zip(a, b)) if not torch.equal(u, v)]
    assert not bad, f"{len(bad)} gradients differ between the schedules"


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("ks,cout,hw,segc", [(3, 80, (20, 24), (16, 16, 16, 16, 16)), (1, 64, (16, 16), (64,)), (3, 48, (40, 36), (16, 32))])
def test_input_gradient_convolution_with_fused_batchnorm_backward_sums(E, L, precision, ks, cout, hw, segc):
    """mdie_conv_desc.bnred: the planar input-gradient convolution that also leaves the per-slab sums of dz and dz * x, folded by
    mdie_bn_bwd_finish -- against the same convolution followed by mdie_bn_bwd_reduce over its output: `da` bit-identical, dgamma /
    dbeta / coef equal to summation-order accuracy (both see the STORED da and x; only sum(dz * xhat) is formed as
    invstd * (sum(dz * x) - mean * sum(dz)) in double instead of term by term).  Shapes: 5 segments with ragged 16x16 tiles and BN = 16
    output tiles, a 1x1 with a 64-wide tile, 8x8-tile territory."""
    import ctypes as C
    dt, td = E.dtype_id(precision), TORCH_DT[precision]
    g = torch.Generator().manual_seed(31)
    B, (H, W), cin = 2, hw, 16
    N = B * H * W
    cl = lambda t: t.cuda().to(td).contiguous(memory_format=torch.channels_last)
    dy = cl(torch.randn(B, cin, H, W, generator=g))
    xs = [cl(torch.randn(B, c, H, W, generator=g) * 1.3 + 0.2) for c in segc]
    w = torch.randn(cout, cin, ks, ks, generator=g) * 0.2
    packed = E.pack_conv_weight(w, dt).cuda()
    ones, zeros = torch.ones(cout, device="cuda"), torch.zeros(cout, device="cuda")
    xcat = torch.cat([t.float() for t in xs], 1)
    mean, var = xcat.mean((0, 2, 3)), xcat.var((0, 2, 3), unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    gamma, beta = torch.rand(cout, generator=g).cuda() + 0.5, torch.randn(cout, generator=g).cuda() * 0.3
    scale = (gamma * invstd).contiguous()
    shift = (beta - mean * scale).contiguous()
    sp = torch.cuda.current_stream().cuda_stream

    def conv(bnred):
        da = torch.full((cout // 16, N, 16), 3.0, dtype=td, device="cuda")
        d = L.ConvDesc()
        d.dtype, d.B, d.H, d.W, d.ksize, d.nseg = dt, B, H, W, ks, 1
        d.inp[0] = L.Seg(dy.data_ptr(), cin, cin)
        d.cin, d.cout = cin, cout
        d.weight, d.post_scale, d.post_shift = packed.data_ptr(), ones.data_ptr(), zeros.data_ptr()
        d.act, d.pool, d.out, d.out_stride, d.out_group_stride = 0, 0, da.data_ptr(), 16, da.stride(0)
        partial = None
        if bnred:
            nslab = L.lib.mdie_conv_bnred_slabs(B, H, W, cout)
            partial = torch.full((nslab, 2, cout), float("nan"), device="cuda")
            r = L.BnReduceFuse()
            r.nseg = len(xs)
            for i, t in enumerate(xs):
                r.x[i] = L.Seg(t.data_ptr(), t.shape[1], t.shape[1])
            r.scale, r.shift, r.partial, r.partial_bytes = scale.data_ptr(), shift.data_ptr(), partial.data_ptr(), partial.numel() * 4
            d.bnred = C.pointer(r)
        L.check(L.lib.mdie_conv_fwd(C.byref(d), sp), "mdie_conv_fwd")
        return da, partial

    da0, _ = conv(False)
    dgb0, coef0 = torch.zeros(2, cout, device="cuda"), torch.zeros(2, cout, device="cuda")
    ws = torch.empty(L.lib.mdie_bn_workspace_bytes(cout), dtype=torch.uint8, device="cuda")
    b = L.BnBwdDesc()
    b.dtype, b.N, b.nseg = dt, N, len(xs)
    for i, t in enumerate(xs):
        b.x[i] = L.Seg(t.data_ptr(), t.shape[1], t.shape[1])
    b.da, b.da_stride, b.da_plane = da0.data_ptr(), 16, N * 16
    b.mean, b.invstd, b.scale, b.shift, b.relu = mean.data_ptr(), invstd.data_ptr(), scale.data_ptr(), shift.data_ptr(), 1
    b.c_real, b.split, b.gap = cout, cout, 0
    b.dgamma, b.dbeta, b.coef = dgb0[0].data_ptr(), dgb0[1].data_ptr(), coef0.data_ptr()
    b.workspace, b.workspace_bytes = ws.data_ptr(), ws.numel()
    L.check(L.lib.mdie_bn_bwd_reduce(C.byref(b), sp), "mdie_bn_bwd_reduce")

    da1, partial = conv(True)
    dgb1, coef1 = torch.zeros(2, cout, device="cuda"), torch.zeros(2, cout, device="cuda")
    f = L.BnBwdFinishDesc()
    f.C, f.N, f.partial, f.n_partial = cout, N, partial.data_ptr(), partial.shape[0]
    f.mean, f.invstd, f.c_real, f.split, f.gap = mean.data_ptr(), invstd.data_ptr(), cout, cout, 0
    f.dgamma, f.dbeta, f.coef = dgb1[0].data_ptr(), dgb1[1].data_ptr(), coef1.data_ptr()
    L.check(L.lib.mdie_bn_bwd_finish(C.byref(f), sp), "mdie_bn_bwd_finish")
    torch.cuda.synchronize()
    assert torch.equal(da0, da1) and not torch.isnan(partial).any()
    tol = 2e-5 if precision == "fp32" else 1e-4
    assert rel_to_max(dgb1, dgb0) <= tol and rel_to_max(coef1, coef0) <= tol and float(dgb0.abs().max()) > 1.0
    # refusals: without the planar output; a partial buffer one slab short; x that does not cover the output channels
    bad = L.ConvDesc()
    C.memmove(C.byref(bad), C.byref(L.ConvDesc()), C.sizeof(L.ConvDesc))
    r = L.BnReduceFuse()
    r.nseg, r.scale, r.shift, r.partial, r.partial_bytes = 1, scale.data_ptr(), shift.data_ptr(), partial.data_ptr(), partial.numel() * 4 - 4 * 2 * cout
    r.x[0] = L.Seg(xs[0].data_ptr(), xs[0].shape[1], xs[0].shape[1])
    for mode in ("not planar", "short", "channels"):
        d = L.ConvDesc()
        d.dtype, d.B, d.H, d.W, d.ksize, d.nseg = dt, B, H, W, ks, 1
        d.inp[0] = L.Seg(dy.data_ptr(), cin, cin)
        d.cin, d.cout = cin, cout
        d.weight, d.post_scale, d.post_shift = packed.data_ptr(), ones.data_ptr(), zeros.data_ptr()
        d.out, d.out_stride, d.out_group_stride = da1.data_ptr(), 16, (0 if mode == "not planar" else da1.stride(0))
        rr = L.BnReduceFuse()
        C.memmove(C.byref(rr), C.byref(r), C.sizeof(L.BnReduceFuse))
        if mode != "channels":
            rr.nseg = len(xs)
            for i, t in enumerate(xs):
                rr.x[i] = L.Seg(t.data_ptr(), t.shape[1], t.shape[1])
        if mode == "channels" or mode == "not planar":
            rr.partial_bytes = partial.numel() * 4
        if mode == "channels" and len(xs) == 1:
            rr.x[0] = L.Seg(xs[0].data_ptr(), 16, xs[0].shape[1])
        d.bnred = C.pointer(rr)
        rc = L.lib.mdie_conv_fwd(C.byref(d), sp)
        assert rc == (-3 if mode == "short" else -1), (mode, rc, L.lib.mdie_last_error())
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_bn_stats_fold_from_partial_sums(L):
    """mdie_bn_stats_fold with x = NULL: the caller supplies per-slab channel sums and sums of squares [n][2][C] (what a producer's
    epilogue would leave); one launch folds them (double precision, slab order) and applies the fold."""
    import ctypes as C
    g = torch.Generator().manual_seed(3)
    Cc, n, per = 32, 37, 50
    y = (torch.randn(n, per, Cc, generator=g) * 2.0 + 0.7).cuda()
    partial = torch.stack((y.sum(1), (y * y).sum(1)), 1).contiguous()           # [n][2][C]
    mv, k = torch.zeros(2, Cc, device="cuda"), torch.zeros(3, Cc, device="cuda")
    gamma, beta = torch.rand(Cc, generator=g).cuda() + 0.5, torch.randn(Cc, generator=g).cuda()
    rm, rv = torch.zeros(Cc, device="cuda"), torch.ones(Cc, device="cuda")
    d = L.BnStatsFoldDesc()
    d.dtype, d.N, d.x, d.C, d.stride = L.F32, n * per, None, Cc, Cc
    d.mean, d.var = mv[0].data_ptr(), mv[1].data_ptr()
    d.workspace, d.workspace_bytes, d.n_partial = partial.data_ptr(), partial.numel() * 4, n
    d.C_fold, d.C_real, d.split, d.gap = Cc, Cc, Cc, 0
    d.fold_mean, d.fold_var, d.gamma, d.beta, d.eps, d.momentum = mv[0].data_ptr(), mv[1].data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-5, 0.1
    d.running_mean, d.running_var = rm.data_ptr(), rv.data_ptr()
    d.scale, d.shift, d.invstd = k[0].data_ptr(), k[1].data_ptr(), k[2].data_ptr()
    L.check(L.lib.mdie_bn_stats_fold(C.byref(d), torch.cuda.current_stream().cuda_stream), "mdie_bn_stats_fold")
    torch.cuda.synchronize()
    flat = y.reshape(-1, Cc).double()
    mean, var = flat.mean(0), flat.var(0, unbiased=False)
    assert rel_to_max(mv[0], mean.float()) <= 1e-6 and rel_to_max(mv[1], var.float()) <= 1e-5
    inv = 1.0 / torch.sqrt(var + 1e-5)
    assert rel_to_max(k[0], (gamma.double() * inv).float()) <= 1e-5 and rel_to_max(k[1], (beta.double() - mean * gamma.double() * inv).float()) <= 1e-5
    assert rel_to_max(rm, (0.1 * mean).float()) <= 1e-5
    d.workspace_bytes = 16
    assert L.lib.mdie_bn_stats_fold(C.byref(d), None) == -1


@pytest.mark.gpu
def test_side_stream_join_survives_a_backward_that_raised(E, monkeypatch):
    """A backward that dies after weight gradients were forked to the side stream never runs its join callback.  The next backward must
    still join (the callback is keyed by the autograd graph task, not by a flag the dead one left set): its optimizer step is compared
    bit for bit with the single-stream schedule."""
    import mdie_amd.train as T
    from models.cdan import CDAN
    from oracle import params as P
    monkeypatch.setattr(T, "WGRAD_STREAM_MIN_PIXELS", 0)
    sd = P.make_state_dict(42)
    x, t = (v.cuda() for v in P.lowlight_batch(9, 4, 128, 128))

    def boom(g):            # a hook on the FIRST layer's weight: fires at the very end of backward, after every weight gradient was launched
        raise RuntimeError("boom")

    def run(side):
        monkeypatch.setattr(T, "WGRAD_STREAM", side)
        torch.manual_seed(5)
        net = CDAN(precision="bf16")
        net.load_state_dict(sd, strict=True)
        net = net.cuda().train()
        opt = torch.optim.Adam(net.parameters(), lr=1e-3, fused=True)
        h = net.encoder.conv1.conv.weight.register_hook(boom)
        with pytest.raises(RuntimeError, match="boom"):
            torch.sqrt((net(x) - t) ** 2 + 1e-6).mean().backward()
        h.remove()
        torch.cuda.synchronize()
        for _ in range(2):
            opt.zero_grad(set_to_none=True)
            loss = torch.sqrt((net(x) - t) ** 2 + 1e-6).mean()
            loss.backward()
            opt.step()
        torch.cuda.synchronize()
        return loss.detach().clone(), [p.detach().clone() for p in net.parameters()]

    a, b = run(True), run(False)
    assert torch.equal(a[0], b[0]) and all(torch.equal(u, v) for u, v in zip(a[1], b[1]))
