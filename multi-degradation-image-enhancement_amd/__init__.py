"""MI355X-native engine for the CDAN/CBAM restoration path (importable as `mdie_amd`).

The product path is HIP only: importing `mdie_amd.lib` fails loudly when
`libmdie_hip.so` has not been built (run `python -c "import __graft_entry__ as g; g.build()"`
or `make -C multi-degradation-image-enhancement_amd/csrc`); there is no CPU fallback.
"""
from .arch import cdan_param_spec  # noqa: F401

__all__ = ["cdan_param_spec"]
