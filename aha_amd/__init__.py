"""aha_amd -- MI355X-native Aha::AC#match (batch Aho-Corasick over a
double-array trie), a drop-in for that one path of chenkovsky/aha.

Layout: csrc/ holds the HIP kernels and the C ABI (include/aha_hip.h);
ac.py mirrors the reference's Aha::AC / Aha::Hit API on top of it.
"""
from .ac import AC, ACBig, ACGroup, AhaError, BitArray, DeviceBuffer, DeviceCorpus, Hit, HIT_DTYPE  # noqa: F401

__all__ = ["AC", "ACBig", "ACGroup", "AhaError", "BitArray", "DeviceBuffer", "DeviceCorpus", "Hit", "HIT_DTYPE"]
