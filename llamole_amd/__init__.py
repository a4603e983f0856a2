import os as _os

# kernel arguments in device memory (the ROCm 7.2 default on this hardware): with host-memory kernargs a chain of queued launches is
# slower than a hipGraph replay (GraphDiT batch 1: 1.27 vs 1.09 ms per step), with device kernargs faster (0.94 ms)
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

