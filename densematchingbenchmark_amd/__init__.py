"""MI355X-native cost-volume -> 3-D aggregation -> disparity-regression path for DenseMatchingBenchmark configs.

``densematchingbenchmark_amd.modeling`` mirrors ``dmb.modeling`` for the stereo hot path (same registries,
constructor kwargs, forward signatures and state_dict keys); ``densematchingbenchmark_amd.ops`` is the functional
layer over the C ABI of ``include/dmb_hip.h`` (libdmb_hip.so, hand-written HIP for gfx950)."""
__version__ = "0.1.0"
