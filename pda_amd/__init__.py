"""pda_amd -- MI355X (gfx950) native implementation of the PDA BPR-MF hot path.

Only the path named in BASELINE.json's north_star lives here: the fused BPR triplet step and the
full-catalogue score + mask + top-K evaluation, behind the reference's MF/ interfaces.
Compute is hand-written HIP in pda_amd/csrc (C ABI: include/pda_hip.h); there is no CPU fallback.
"""
__version__ = "0.1.0"
