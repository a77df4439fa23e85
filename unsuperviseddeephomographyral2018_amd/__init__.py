"""MI355X-native hot path of "Unsupervised Deep Homography" (RA-L 2018):
Tensor-DLT + projective bilinear warp + photometric L1, forward and backward, as hand-written HIP
kernels for gfx950 behind the C ABI in include/uh_hotpath.h; Python here is only the host-side
mirror of the reference's operator interface (homography_model.py, utils/tf_spatial_transformer.py).
"""
__version__ = '0.1.0'
