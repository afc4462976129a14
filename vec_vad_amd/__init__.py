"""vec_vad_amd -- MI355X (gfx950) implementation of VEC_VAD's spatio-temporal cube completion hot path.

Layout: csrc/ (hand-written HIP kernels + C ABI, built into csrc/libvecvad_hip.so), _lib.py (ctypes binding),
bank.py (grouped UNet-bank engine), unet.py (reference module surface), flow_ops.py (FlowNet2 native ops).
"""
__version__ = '0.1.0'
