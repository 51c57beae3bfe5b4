"""CPU oracle for the segmentation hot path — TEST INFRASTRUCTURE ONLY.

This package restates, on the CPU, the tf.keras (Keras-2, TF 2.x) semantics that the reference
(A511-1103/building-detection) relies on for its DeepLabv3+/ASPP, DeepLab-BAM, SCSE-UNet, Res34-UNet and
HRNet forward+backward path.  TensorFlow is not installed in the authoring container nor on the GPU box and
the reference ships no tests, goldens or weights, so

    PARITY UNPINNED: nothing in this oracle could be checked against an execution of the reference or a
    reference-held golden vector.  The only reference known-answer (22,910,272 Res34 backbone parameters,
    train_model/res34.py:305) is reproduced.  What pins the oracle instead: two independent CPU
    implementations of every primitive (torch-functional in `tfops.py`, plain numpy/C in `ops_np.py` /
    `conv_ref.c`) that must agree, plus fp64 finite-difference gradient checks (tests/test_oracle_*.py).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this package, and
only as the checker.  The product (`building_detection_amd`) never imports it and has no CPU fallback.
"""
