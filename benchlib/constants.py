"""Algorithmic work of the BASELINE configs and the roofs they are priced against."""
VOXELS = 96 * 48 * 96
# algorithmic work per chunk
BACKBONE = dict(bytes=201.6e6, flops=17.72e9)
RPN = dict(bytes=59.8e6, flops=24.86e9)
ALGO = {
    "backbone_rpn": dict(bytes=261.4e6, flops=42.58e9),
    "detect": dict(bytes=261.4e6 + 2 * 3.54e6 + 200 * 32768 + 8.4e6, flops=42.58e9 + 0.9e9),
    "images": dict(bytes=517e6 + 28.3e6 + 59.8e6 + 226.5e6, flops=29.1e9 + 24.86e9),
    "scene": dict(bytes=261.4e6 + 2 * 3.54e6 + 200 * 32768 + 8.4e6, flops=42.58e9 + 0.9e9),
}
DOMINANT_FLOPS = 2.0 * 6912 * 256 * 128 * 27        # rpn_net_level{1,2}: 12.23 GFLOP per launch (ALGORITHMIC = direct-convolution count)
WINOGRAD_REDUCTION = 27 * 8 / 64.0                  # F(2x2x2, 3x3x3): 64 products per 2x2x2 output block instead of 216
FP32_PEAK_TF = 157.3
HBM_PEAK_GBS = 8000.0
WORKLOAD_TEXT = {
    "backbone_rpn": "config[1]: one 96x48x96 chunk per pipeline, geometry-only, HIP 3D-conv backbone + RPN (convs, heads, "
                    "softmax), weights seeded synthetic",
    "detect": "config[2]: backbone + RPN + decode/sort/NMS + RoI pooling + classifier",
    "images": "config[3]: 5-view back-projection gather + colour/geometry backbone + RPN",
    "scene": "config[4]: 32-chunk scene sharded chunk->rank, per-chunk detection, one RCCL all-gather of record blocks, "
             "whole-scene 3D NMS on every rank",
}
