"""foundationpose_amd: MI355X-native (gfx950) render-and-compare hot path of FoundationPose.

The compute path is the hand-written HIP library `foundationpose_amd/lib/libfoundationpose_amd.so`
(C-ABI in include/foundationpose_amd.h).  There is no CPU fallback: importing the ops without the
built library raises.
"""
__version__ = '0.1.0'
