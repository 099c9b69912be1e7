"""MI355X-native Smith-Waterman / PairHMM hot paths: ctypes view of the C-ABI (include/agx.h)."""
