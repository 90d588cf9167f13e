"""efficientq_amd -- MI355X-native layer-wise PTQ calibration (the hot path of rongzhao-zhang/EfficientQ).

    csrc/            hand-written HIP kernels + the C ABI (include/effq_hip.h) -> libeffq_hip.so
    _lib, hip_ops    ctypes binding and tensor-level front end (no CPU fallback)
    qconv            PTQConv / EfficientQConvHIP (reference class contract)
    unet             UResQ host graph (reference module tree / state_dict keys)
    calibrate        do_ptq orchestrator, BN folding, attention-mask pyramid
    config           CLI / YAML surface, factories, the shipped BraTS / LiTS net definitions
    entrance         `python -m efficientq_amd.entrance ptq ...`
    mixed            per-layer mixed-precision search harness
    synth            seeded synthetic volumes / random-init networks
"""
__version__ = "0.1.0"
