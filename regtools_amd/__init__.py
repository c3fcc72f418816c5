"""regtools_amd -- MI355X (gfx950) implementation of the `regtools junctions extract` hot path.

Layout: csrc/ (HIP kernels, C ABI, host CLI, synthetic-input tooling), _ffi.py (ctypes over the C ABI),
extractor.py (host mirror of the reference's JunctionsExtractor interface), cse.py (mirrors of CisSpliceEffectsIdentifier /
CisSpliceEffectsAssociator / VariantsAnnotator / JunctionsAnnotator), synth.py (synthetic inputs), distributed.py (shard merge).
"""
from .extractor import Context, Junction, JunctionsExtractor, PinnedBuffer, Pipeline, RegtoolsError, extract_multi, junctions_extract  # noqa: F401
from .cse import (CisSpliceEffectsAssociator, CisSpliceEffectsIdentifier, JunctionsAnnotator, VariantsAnnotator,  # noqa: F401
                  cis_splice_effects_identify)

__version__ = "0.1"
