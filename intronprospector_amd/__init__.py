"""MI355X-native junction extraction (the hot path of diekhans/intronProspector).

Importing the package loads libipx.so (built in-tree from csrc/ with hipcc for
gfx950).  There is no CPU fallback: a missing library raises ImportError here,
and a missing GPU raises when an extractor is created.
"""
from . import _lib

_lib.load()

from .extractor import (JunctionsExtractor, JunctionsExtractorGroup, IpxError, expand_histogram,  # noqa: E402,F401
                        offsets_per_junction)
