"""Drop-in for the reference's ``model/unet.py`` import path (``from model.unet import SelfCompleteNet4, ...``,
train.py:12 / test.py:11).  The implementation lives in vec_vad_amd/unet.py (HIP UNet-bank engine)."""
from vec_vad_amd.unet import (double_conv, inconv, down, up, outconv,  # noqa: F401
                              SelfCompleteNet4, SelfCompleteNetFull, SelfCompleteNet1raw1of)
