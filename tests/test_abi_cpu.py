"""CPU-side checks of the C ABI: the library loads without a GPU, exports every function
include/mdie.h declares, validates arguments before touching the device, and its host-side
packers produce the documented layouts."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

import mdie_amd.engine as E
import mdie_amd.lib as L
from oracle import params as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_declared_symbol_is_exported_and_bound():
    text = open(os.path.join(ROOT, "include", "mdie.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = set(re.findall(r"\b(mdie_[a-z0-9_]+)\s*\(", text))
    assert declared, "no declarations parsed"
    assert declared == set(L.SIGNATURES), declared ^ set(L.SIGNATURES)
    raw = C.CDLL(L.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), name
    assert L.lib.mdie_abi_version() == L.ABI_VERSION


def bf16_bits(a):
    u = a.astype(np.float32).view(np.uint32).astype(np.uint64)
    return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) & 0xFFFF).astype(np.uint16)


@pytest.mark.parametrize("dtype", [L.F32, L.BF16])
@pytest.mark.parametrize("transposed", [False, True])
def test_pack_conv_weight_layout(dtype, transposed):
    rng = np.random.default_rng(0)
    cout, cin, ks = 19, 35, 3          # final_dense-like: 3 + 2*16 real input channels, base stored in 16
    split, gap = 3, 13
    cin_st, cout_st = 48, 32
    w = rng.standard_normal((cin, cout, ks, ks) if transposed else (cout, cin, ks, ks)).astype(np.float32)
    packed = E.pack_conv_weight(torch.from_numpy(w), dtype, transposed=transposed, cout_stored=cout_st,
                                cin_stored=cin_st, split=split, gap=gap)
    kc = 16 if dtype == L.F32 else 32
    nchunk = -(-cin_st // kc)
    assert packed.numel() == nchunk * 9 * cout_st * 64 == L.lib.mdie_conv_weight_bytes(dtype, ks, cin_st, cout_st)
    vec = kc // 4
    expect = np.zeros((nchunk, 4, 9, cout_st, vec), np.float32)
    for o in range(cout):
        for c in range(cin):
            cs = c + (gap if c >= split else 0)
            for kh in range(3):
                for kw in range(3):
                    v = w[c, o, 2 - kh, 2 - kw] if transposed else w[o, c, kh, kw]
                    k = cs % kc
                    expect[cs // kc, k // vec, kh * 3 + kw, o, k % vec] = v
    if dtype == L.F32:
        got = packed.numpy().view(np.float32).reshape(expect.shape)
        assert np.array_equal(got, expect)
    else:
        got = packed.numpy().view(np.uint16).reshape(expect.shape)
        assert np.array_equal(got, bf16_bits(expect))


def test_pack_params_roundtrip_and_errors():
    sd = P.make_state_dict(42)
    for dt in (L.F32, L.BF16):
        blob = E.pack_checkpoint(sd, dt)
        assert blob.numel() == L.lib.mdie_cdan_param_bytes(dt)
        assert torch.equal(blob, E.pack_checkpoint(sd, dt))
    broken = dict(sd)
    del broken["decoder.cbam2.ChannelGate.mlp.3.bias"]
    with pytest.raises(L.MdieError, match="decoder.cbam2.ChannelGate.mlp.3.bias"):
        E.pack_checkpoint(broken, L.F32)
    wrong = dict(sd)
    wrong["encoder.conv2.conv.weight"] = torch.zeros(128, 64, 3, 2)
    with pytest.raises(L.MdieError, match="encoder.conv2.conv.weight"):
        E.pack_checkpoint(wrong, L.BF16)


def test_folded_batchnorm_in_blob():
    """encoder.conv1 is the first blob entry: packed weight, then post_scale, post_shift."""
    sd = P.make_state_dict(42)
    blob = E.pack_checkpoint(sd, L.F32).numpy()
    wbytes = L.lib.mdie_conv_first_weight_bytes(L.F32, 64)
    scale = blob[wbytes:wbytes + 256].view(np.float32)
    shift = blob[wbytes + 256:wbytes + 512].view(np.float32)
    g, b = sd["encoder.conv1.bn.weight"].double(), sd["encoder.conv1.bn.bias"].double()
    m, v = sd["encoder.conv1.bn.running_mean"].double(), sd["encoder.conv1.bn.running_var"].double()
    s = g / torch.sqrt(v + 1e-5)
    t = (sd["encoder.conv1.conv.bias"].double() - m) * s + b
    assert np.allclose(scale, s.numpy(), rtol=1e-6)
    assert np.allclose(shift, t.numpy(), rtol=1e-5, atol=1e-6)


def test_sizes_and_work_model():
    assert L.lib.mdie_cdan_workspace_bytes(L.BF16, 1, 36, 32) == 0          # H not a multiple of 8
    assert L.lib.mdie_cdan_workspace_bytes(L.BF16, 0, 32, 32) == 0
    a = L.lib.mdie_cdan_workspace_bytes(L.BF16, 2, 64, 64)
    b = L.lib.mdie_cdan_workspace_bytes(L.F32, 2, 64, 64)
    assert 0 < a < b
    # SURVEY.md section 6/8d constants
    assert L.lib.mdie_cdan_flops(1, 256, 256) == pytest.approx(16570124288.0)
    act = L.lib.mdie_cdan_algorithmic_bytes(1, 256, 256, 2) - 2 * 3585663
    assert act / 2 == pytest.approx(52.39e6, rel=1e-3)


def test_argument_validation_needs_no_gpu():
    d = L.CdanFwdDesc()
    d.dtype, d.B, d.H, d.W = L.BF16, 1, 36, 32
    assert L.lib.mdie_cdan_forward(C.byref(d), None) == -1
    assert b"multiples of 8" in L.lib.mdie_last_error()
    c = L.ConvDesc()
    c.dtype, c.B, c.H, c.W, c.ksize, c.nseg = 7, 1, 8, 8, 3, 1
    assert L.lib.mdie_conv_fwd(C.byref(c), None) == -1
    assert b"dtype" in L.lib.mdie_last_error()
    assert L.lib.mdie_upsample2x_add(L.F32, 1, 4, 4, 12, 16, 16, 16, 16, 16, 16, None) == -1
    k = L.CbamDesc()
    k.dtype, k.B, k.H, k.W, k.C = L.F32, 1, 4, 4, 48
    assert L.lib.mdie_cbam_fwd(C.byref(k), None) == -1
    assert b"power of two" in L.lib.mdie_last_error()


def test_module_front_end_contract():
    from models.cbam import CBAM
    from models.cdan import CDAN
    m = CDAN()
    assert len(m.state_dict()) == 236
    assert sum(p.numel() for p in m.parameters()) == 3585663
    m.load_state_dict(P.make_state_dict(7), strict=True)
    with pytest.raises(L.MdieError, match="no CPU fallback"):
        m.eval()(torch.rand(1, 3, 32, 32))
    c = CBAM(64)
    assert list(c.state_dict())[0] == "ChannelGate.mlp.1.weight"
    with pytest.raises(NotImplementedError):
        CBAM(64, reduction_ratio=8)


@pytest.mark.parametrize("dtype", [L.F32, L.BF16])
def test_pack_first_layer_weight(dtype):
    rng = np.random.default_rng(1)
    w = rng.standard_normal((64, 3, 3, 3)).astype(np.float32)
    n = L.lib.mdie_conv_first_weight_bytes(dtype, 64)
    dst = torch.zeros(n, dtype=torch.uint8)
    L.check(L.lib.mdie_pack_conv_first_weight(dtype, w.ctypes.data, 64, 64, dst.data_ptr()), "pack")
    im2col = np.zeros((64, 32), np.float32)
    for k in range(27):
        im2col[:, k] = w[:, k % 3, (k // 3) // 3, (k // 3) % 3]
    if dtype == L.F32:
        got = dst.numpy().view(np.float32).reshape(2, 64, 16)
        assert np.array_equal(np.concatenate((got[0], got[1]), axis=1), im2col)
    else:   # 16-bit: k' = tap*4 + c, 32 per step (taps 0..7 | tap 8), every other element zero
        pix = np.zeros((64, 64), np.float32)
        for tap in range(9):
            for c in range(3):
                pix[:, tap * 4 + c] = w[:, c, tap // 3, tap % 3]
        got = dst.numpy().view(np.uint16).reshape(2, 64, 32)
        assert np.array_equal(np.concatenate((got[0], got[1]), axis=1), bf16_bits(pix))


def test_custom_torch_ops_are_registered_and_have_no_cpu_kernel():
    """torch.ops.mdie.* (ops.py): schema visible, fake implementation traces shapes, a CPU tensor fails in the dispatcher"""
    import pytest
    import torch
    import mdie_amd.ops  # noqa: F401
    assert "cdan_forward" in str(torch.ops.mdie.cdan_forward.default._schema)
    with torch._subclasses.fake_tensor.FakeTensorMode():
        x = torch.empty(2, 3, 64, 64)
        y = torch.ops.mdie.cdan_forward(x, torch.empty(8, dtype=torch.uint8), torch.empty(8, dtype=torch.uint8), 1, 0, 0)
        assert y.shape == (2, 3, 64, 64) and y.dtype == torch.float32
    with pytest.raises(NotImplementedError):
        torch.ops.mdie.cdan_forward(torch.zeros(1, 3, 8, 8), torch.zeros(8, dtype=torch.uint8), torch.zeros(8, dtype=torch.uint8), 1, 0, 0)
    with pytest.raises(NotImplementedError):
        torch.ops.mdie.psnr_ssim(torch.zeros(1, 3, 16, 16), torch.zeros(1, 3, 16, 16))


def test_shipped_code_objects_pass_the_isa_guard():
    """tools/isa_guard.py on the library the tests load: NO kernel reads a VGPR pair through the op_sel / op_sel_hi modifiers of a
    packed-f32 instruction -- the form that returned wrong values with MFMAs in flight on the CU, inside one kernel in round 2 and
    across kernels (MFMA-free CBAM backward next to the weight-gradient kernels on a second stream) in round 3, profiles/LEDGER.md (rounds 1-4) section 4
    finding 6 -- and no kernel of the library spills registers to scratch."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_guard
    rows = isa_guard.audit(L.LIB_PATH)
    assert len(rows) > 100 and sum(1 for r in rows if r["mfma"]) > 50, "disassembly found too few kernels: the audit itself is broken"
    assert sum(r["pk"] for r in rows) > 100, "the audit no longer sees packed-f32 instructions at all (the convolutions' plain packed epilogues)"
    assert isa_guard.vgpr_swizzle("v_pk_fma_f32 v[2:3], v[4:5], v[6:7], v[8:9] op_sel:[0,1,1] op_sel_hi:[1,0,0]"), "the audit no longer recognises the forms it exists to find"
    assert not isa_guard.vgpr_swizzle("v_pk_fma_f32 v[2:3], v[4:5], v[6:7], v[8:9]") and not isa_guard.vgpr_swizzle("v_pk_mul_f32 v[2:3], v[4:5], 1.0 op_sel_hi:[1,0]")
    bad = isa_guard.violations(rows)
    assert not bad, [(r["pretty"], r["pk_sel"][:2]) for r in bad]
    spills = [(r["pretty"], r["scratch"], r["scratch_hot"]) for r in isa_guard.spills(rows)]     # (one documented exception: isa_guard.spills)
    assert not spills, spills


def test_numa_binding_helper_is_inert_without_a_gpu(monkeypatch):
    """host.bind_to_gpu_numa: the CPU-list parser, and that placement never raises or changes anything where there is no GPU / sysfs entry"""
    import os
    from mdie_amd import host as H
    assert H._cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11} and H._cpulist("") == set()
    before = os.sched_getaffinity(0)
    assert H.bind_to_gpu_numa(0) is None and os.sched_getaffinity(0) == before
    monkeypatch.setenv("MDIE_NUMA_BIND", "0")
    assert H.bind_to_gpu_numa(0) is None
    n = H.cpu_share()
    assert 1 <= n <= len(os.sched_getaffinity(0))
    import torch
    assert H.cap_cpu_threads() <= max(n, 1) or torch.get_num_threads() <= n


def test_ctypes_descriptors_match_the_header_layout():
    """lib.py mirrors include/mdie.h by hand: a field appended in one and not in the other shifts everything behind it.  The structs the
    engine passes most are checked against a C compiler's view of the header (gcc is in the image; no GPU involved)."""
    import subprocess
    import tempfile
    import mdie_amd.lib as L
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "mdie.h"\nint main(void) {\n'
    checks = {"mdie_conv_desc": (L.ConvDesc, ["dtype", "ksize", "cin", "weight", "out", "tr", "out_group_stride", "bnred", "blob_delta", "share_cu"]),
              "mdie_seg": (L.Seg, ["ptr", "channels", "stride"]),
              "mdie_tr_fuse": (L.TrFuse, ["weight", "c0", "partial_out", "act", "out_nchw3"])}
    for name, (_, fields) in checks.items():
        src += f'  printf("{name} %zu", sizeof({name}));\n'
        for f in fields:
            src += f'  printf(" %zu", offsetof({name}, {f}));\n'
        src += '  printf("\\n");\n'
    src += "  return 0;\n}\n"
    with tempfile.TemporaryDirectory() as d:
        c, exe = os.path.join(d, "layout.c"), os.path.join(d, "layout")
        with open(c, "w") as fh:
            fh.write(src)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe], check=True)
        out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout
    import ctypes as C
    for line in out.splitlines():
        name, size, *offs = line.split()
        cls, fields = checks[name]
        assert C.sizeof(cls) == int(size), (name, C.sizeof(cls), size)
        for f, o in zip(fields, offs):
            py = {"in": "inp"}.get(f, f)
            assert getattr(cls, py).offset == int(o), (name, f, getattr(cls, py).offset, o)


def test_conv4_kernel_choice_is_only_timed_where_conv_wide_applies():
    import mdie_amd.engine as E
    import mdie_amd.lib as L
    assert E._share_cu_eligible(L.BF16, 32, 256, 256) and E._share_cu_eligible(L.F16, 8, 256, 256) and E._share_cu_eligible(L.BF16, 4, 1024, 1024)
    assert not E._share_cu_eligible(L.F32, 32, 256, 256)          # fp32 never takes conv_wide
    assert not E._share_cu_eligible(L.BF16, 2, 64, 64)            # the 8x8 map is no whole 32x16 tile
    assert not E._share_cu_eligible(L.BF16, 1, 256, 256)          # 16 items: conv_wide declines below 96
