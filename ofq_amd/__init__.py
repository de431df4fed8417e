"""ofq_amd — MI355X-native (gfx950) hot path of OFQ quantisation-aware training.

Host side is PyTorch-ROCm (modules, autograd at module granularity, optimizer, torch.distributed over
RCCL); every fake-quant op is a hand-written HIP kernel behind the C ABI in include/ofq_hip.h.
"""
__version__ = "0.1.0"
