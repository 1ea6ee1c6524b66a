"""ctypes handle on oracle/_ref/libpfdsp_ref.so — the REAL reference mixers (src/pf_mixer.cpp of
marton78/pffft, SSE variants included) compiled from their own source by oracle/Makefile.

TEST INFRASTRUCTURE ONLY (tests/test_pfdsp*.py): the checker for SURVEY.md §8 row f-4.  The product
(pffft_amd.pfdsp, libpfdsp_hip.so) never loads it.  Both libraries export the same names, so each is
opened through its own RTLD_LOCAL handle; the typed binding (struct layouts of
include/pffft/pf_mixer.h:61-280) is shared with the product's ctypes layer, pffft_amd.pfdsp.MixerABI.
"""
from __future__ import annotations

import os

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_SO = os.path.join(_HERE, "_ref", "libpfdsp_ref.so")
_ABI = None


def available() -> bool:
    return os.path.exists(REF_SO)


def get():
    global _ABI
    if _ABI is None:
        from pffft_amd.pfdsp import MixerABI
        _ABI = MixerABI(REF_SO)
    return _ABI
