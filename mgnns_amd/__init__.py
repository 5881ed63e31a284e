"""MI355X-native forward hot path of MGNNS (text / object / scene GCN channels +
single-query multi-head fusion) behind the reference's nn.Module surface.

Nothing here computes on the CPU: every operator is a hand-written gfx950 kernel
reached through the C ABI declared in include/mgnns_hip.h; a missing library or a
CPU tensor is an error, not a fallback.
"""
__version__ = "0.1.0"
