"""MI355X-native engine for the DeepfakeDetection image-classifier hot loop.

Layout: csrc/ (HIP kernels + the C ABI of include/dfd_hip.h), kernels.py (tensor front
end), engine/modules (the HIP-backed nn.Modules) and the host-side mirror of the
reference's plug-in surface (orchestration/, trainers/).
"""

__version__ = "0.1.0"
