#!/usr/bin/env python3
"""Static audit of the gfx950 code objects inside the SHIPPED libmdie_hip.so (no GPU needed).

    python tools/isa_guard.py [path/to/libmdie_hip.so] [--list]

For every kernel: number of v_mfma, of packed-f32 vector instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32), how
many of those read a VGPR source pair through an `op_sel` / `op_sel_hi` modifier (the SPLAT / SWAP forms: one 32-bit half of
a register pair feeds both results, or the halves are exchanged), scratch instructions (register spills) and the code object it lives in.

Why.  Round 2 found `v_pk_fma_f32 ... op_sel:[0,1,1]` / `op_sel_hi:[1,0,0]` returning wrong values in lanes 48..63 in the
epilogue of conv_first_pool_kernel -- a kernel that also issues MFMAs -- while another kernel shared the CU; the ISA of the
failing and the passing build is under profiles/r03_pkfma_*.txt and the analysis in profiles/LEDGER.md (rounds 1-4) section 4 (finding 6): every
documented MFMA -> VALU wait-state rule is met with margin in both builds, so the mechanism is NOT explained.  Round 3 then saw
the same thing ACROSS kernels: the MFMA-free backward kernels of cbam_train.hip (10-49 such forms each), bit-reproducible on
one stream, gave different gradients in one training step out of three once the weight-gradient MFMA kernels ran beside them on
a second stream -- and never (0 of 12 rounds, tools/race_hunt_train.py) once that file was built without packed f32.  The rule
this tool enforces (tests/test_abi_cpu.py runs it on every build) is therefore library-wide:

    no kernel contains a packed-f32 instruction that reads a VGPR pair through op_sel / op_sel_hi,

so that a compiler upgrade, an SLP-vectoriser decision or a new epilogue cannot bring the form back unnoticed.  The MFMA files
(conv*.hip) keep plain packed f32 (distinct register pairs, no swizzle) in their epilogues and pre-activations; every other
file is built with the packed-f32 feature off (csrc/Makefile, NOPK).
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
PK = ("v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32")


def demangle(names):
    try:
        out = subprocess.run([os.path.join(LLVM, "llvm-cxxfilt")] if os.path.exists(os.path.join(LLVM, "llvm-cxxfilt")) else ["c++filt"],
                             input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def vgpr_swizzle(text):
    """True if a packed-f32 instruction reads a VGPR source pair through a non-default half selection: op_sel (which half
    feeds the LOW result, default 0 = low) / op_sel_hi (which half feeds the HIGH result, default 1 = high).  Splat
    (0,0) / (1,1) and swap (1,0) forms count; the same modifiers on an inline constant (`1.0 op_sel_hi:[1,0]`: how an f32
    immediate is given to both halves) do not -- no register is read through them."""
    m = re.match(r"(\S+)\s+(.*)$", text)
    ops = [o.strip() for o in re.split(r",\s*(?![^\[]*\])", re.sub(r"\s+op_sel(_hi)?:\[[^\]]*\]", "", m.group(2)))]
    srcs = ops[1:]                                   # ops[0] = vdst
    sel = re.search(r"op_sel:\[([^\]]*)\]", text)
    sel_hi = re.search(r"op_sel_hi:\[([^\]]*)\]", text)
    lo = [int(v) for v in sel.group(1).split(",")] if sel else [0] * len(srcs)
    hi = [int(v) for v in sel_hi.group(1).split(",")] if sel_hi else [1] * len(srcs)
    for i, src in enumerate(srcs):
        if src.startswith(("v[", "a[")) and i < len(lo) and i < len(hi) and (lo[i], hi[i]) != (0, 1):
            return True
    return False


def audit(so_path):
    """-> list of dicts, one per kernel: name, object, mfma, pk, pk_sel (list of instruction texts), scratch"""
    so_path = os.path.abspath(so_path)
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, "lib.so")
        os.symlink(so_path, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], cwd=tmp, check=True, capture_output=True)
        objs = sorted(f for f in os.listdir(tmp) if "amdgcn" in f and "gfx950" in f)
        if not objs:
            raise RuntimeError(f"no gfx950 code object found in {so_path}")
        for obj in objs:
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", os.path.join(tmp, obj)],
                                 check=True, capture_output=True, text=True).stdout
            cur = None
            for line in dis.split("\n"):
                m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
                if m:
                    cur = {"name": m.group(1), "object": obj, "mfma": 0, "pk": 0, "pk_sel": [], "scratch": 0, "scratch_hot": 0, "insts": 0}
                    rows.append(cur)
                    continue
                if cur is None:
                    continue
                t = line.strip()
                if not t or t.startswith(("//", ";")):
                    continue
                t = re.sub(r"\s*//.*$", "", t)
                op = t.split()[0]
                cur["insts"] += 1
                if op.startswith("v_mfma") or op.startswith("v_smfma"):
                    cur["mfma"] += 1
                elif op in PK:
                    cur["pk"] += 1
                    if vgpr_swizzle(t):
                        cur["pk_sel"].append(t)
                elif op.startswith("scratch_"):
                    cur["scratch"] += 1
                    if cur["mfma"]:
                        cur["scratch_hot"] += 1        # behind the kernel's first MFMA in program order: inside its main loop
    rows = [r for r in rows if r["insts"] > 4]          # (drop padding stubs)
    dm = demangle([r["name"] for r in rows])
    for r in rows:
        r["pretty"] = dm.get(r["name"], r["name"])
    return rows


def violations(rows):
    return [r for r in rows if r["pk_sel"]]


def spills(rows):
    """kernels that touch scratch and may not.  One family may, OUTSIDE its main loop: the several-weight-sets form of conv_thin_kernel
    (last template argument true) keeps 72 weight registers, 48 prefetch registers and ~35 lane constants live across a loop over
    stretches of its run and spills 28-38 registers in the per-stretch prologue -- never behind its first MFMA."""
    out = []
    for r in rows:
        if not r["scratch"]:
            continue
        multi_thin = ("conv_thin_kernel<" in r["pretty"] and r["pretty"].split(">(")[0].rstrip().endswith("true")) or \
            re.search(r"conv_thin_kernelI.*Lb1EEEvNS_8ThinArgsEi$", r["name"]) is not None     # (bf16 / f16 names do not always demangle)
        if multi_thin and r["scratch_hot"] == 0:
            continue
        out.append(r)
    return out


def main(argv):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = [a for a in argv if not a.startswith("--")]
    so = args[0] if args else os.path.join(root, "multi-degradation-image-enhancement_amd", "libmdie_hip.so")
    rows = audit(so)
    if "--list" in argv:
        print(f"# {len(rows)} kernels in {os.path.basename(so)}: v_mfma | packed f32 | of those with op_sel | scratch | kernel")
        for r in sorted(rows, key=lambda r: (-len(r["pk_sel"]), -r["mfma"])):
            if r["mfma"] or r["pk"] or r["scratch"]:
                print(f"{r['mfma']:6d} {r['pk']:5d} {len(r['pk_sel']):5d} {r['scratch']:5d}  {r['pretty'][:150]}")
    bad = violations(rows)
    print(f"{len(rows)} kernels; {sum(1 for r in rows if r['mfma'])} issue MFMAs; {sum(r['pk'] for r in rows)} packed-f32 instructions in "
          f"{sum(1 for r in rows if r['pk'])} kernels, of which read a VGPR pair through op_sel / op_sel_hi: {sum(len(r['pk_sel']) for r in bad)} in {len(bad)} kernels; "
          f"kernels touching scratch: {sum(1 for r in rows if r['scratch'])} (not allowed: {len(spills(rows))})")
    for r in bad:
        print(f"VIOLATION: {r['pretty'][:160]}: {len(r['pk_sel'])} packed-f32 op_sel forms ({r['mfma']} v_mfma), e.g. {r['pk_sel'][0]}")
    for r in spills(rows):
        print(f"SPILL: {r['pretty'][:160]}: {r['scratch']} scratch instructions, {r['scratch_hot']} behind the first MFMA")
    return 1 if (bad or spills(rows)) else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
