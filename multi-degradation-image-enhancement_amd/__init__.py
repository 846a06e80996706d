"""MI355X-native engine for the CDAN/CBAM restoration path (importable as `mdie_amd`).

The product path is HIP only: importing `mdie_amd.lib` fails loudly when
`libmdie_hip.so` has not been built (run `python -c "import __graft_entry__ as g; g.build()"`
or `make -C multi-degradation-image-enhancement_amd/csrc`); there is no CPU fallback.
"""
import os as _os

# Kernel arguments in DEVICE memory (HIP_FORCE_DEV_KERNARG=1; the runtime's default keeps them in host memory, read over PCIe by the first
# waves of every launch).  This engine's launches carry descriptors of a few hundred bytes and a forward is 39 of them back to back:
# measured on one MI355X box, alternating processes, 28.8 -> 30.8 k images/s eager, 29.0 -> 30.9 k as a hipGraph, routed 26.4 -> 28.3 k
# (profiles/LEDGER.md (rounds 1-4) section 5, profiles/r04r_dev_kernarg_ab.txt).  The HIP runtime reads the variable when it initialises -- on the first HIP
# call, not at import -- so a default set here takes effect as long as the package is imported before the process touches the GPU
# (bench.py, run.py, tests/conftest.py and __graft_entry__ also set it first thing); an explicit setting in the environment wins.
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

from .arch import cdan_param_spec  # noqa: F401,E402

__all__ = ["cdan_param_spec"]
