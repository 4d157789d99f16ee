"""scone_gcn_amd: MI355X-native (gfx950) hot path of SCoNe -- the Hodge-Laplacian convolution layers, their
backward and the train step -- behind the reference's scone_func / Scone_GCN call surface.

Importing the package does not need a GPU; calling any model function does (there is no CPU fallback).
"""
from .complex import SimplicialComplex, Shift, Bconds          # noqa: F401
from .synthetic_data_gen import SparseFlows                    # noqa: F401

__all__ = ["SimplicialComplex", "Shift", "Bconds", "SparseFlows"]
