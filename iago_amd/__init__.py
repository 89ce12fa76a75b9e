"""iago_amd -- MI355X-native hot path of shionhonda/IaGo (8x8 Othello PV-MCTS
self-play): batched bitboard HIP kernels behind a C ABI (include/iago_hip.h),
PyTorch-ROCm policy/value nets, and host-side mirrors of the reference's Python
interfaces.  See DESIGN.md.
"""
__version__ = "0.1.0"
