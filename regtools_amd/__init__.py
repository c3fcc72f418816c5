"""regtools_amd -- MI355X (gfx950) implementation of the `regtools junctions extract` hot path.

Layout: csrc/ (HIP kernels, C ABI, host CLI, synthetic-input tooling), _ffi.py (ctypes over the C ABI),
extractor.py (host mirror of the reference's JunctionsExtractor interface), synth.py (synthetic BAM/BAI).
"""
from .extractor import Context, Junction, JunctionsExtractor, RegtoolsError, junctions_extract  # noqa: F401
from .cse import CisSpliceEffectsIdentifier, cis_splice_effects_identify  # noqa: F401

__version__ = "0.1"
