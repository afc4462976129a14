"""Import-compatible front of the reference's ``FlowNet2_src`` package (``from FlowNet2_src import FlowNet2``,
calc_optical_flow.py:7): re-exports the gfx950 implementation in ``vec_vad_amd``.  The training-side pieces of the
reference package (losses, datasets, main.py) and ``flow_to_image`` (only referenced from commented-out visualisation,
calc_optical_flow.py:78-80) are not part of the flow-extraction path and are not provided."""
from vec_vad_amd.flownet2 import FlowNet2, FlowNetC, FlowNetS, FlowNetSD, FlowNetFusion  # noqa: F401
from vec_vad_amd.flow_ops import Correlation, Resample2d, ChannelNorm  # noqa: F401
