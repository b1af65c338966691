"""sparsernns_amd -- MI355X-native fixed-point S5 inference path behind the call surface of
stevenabreu7/SparseRNNs' ``fxprun.py`` / ``fxpmodel.py`` / ``fxparray.py``.

Submodules: ``fxparray`` (FxpArray ops on the GPU), ``fxpmodel`` (model classes), ``engine`` (fused
forward over the C ABI), ``synth`` (synthetic NDNS-shaped models), ``_lib`` (ctypes binding of
libs5fxp.so; importing it fails loudly when the HIP extension has not been built).
"""
__version__ = "0.1.0"
