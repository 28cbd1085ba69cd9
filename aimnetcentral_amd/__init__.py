"""aimnetcentral_amd - MI355X-native AIMNet2 inference engine (HIP/gfx950) behind the
AIMNet2Calculator / AIMNet2ASE API of isayevlab/aimnetcentral.

Only the energy/force/virial hot path lives here: csrc/ holds the HIP kernels and
the C ABI (include/aimnet_hip.h), the Python modules are the host-side mirror of the reference
interface.  Importing the package never loads the GPU library; constructing a calculator does,
and fails loudly (HipLibraryError) when it is missing - there is no CPU or PyTorch fallback.
"""
from ._lib import HipLibraryError

__all__ = ["AIMNet2Calculator", "AIMNet2ASE", "AIMNet2TorchSim", "HipEngine", "HipLibraryError", "load_model"]


def __getattr__(name):
    if name == "AIMNet2Calculator":
        from .calculator import AIMNet2Calculator

        return AIMNet2Calculator
    if name == "AIMNet2ASE":
        from .aimnet2ase import AIMNet2ASE

        return AIMNet2ASE
    if name == "AIMNet2TorchSim":
        from .aimnet2torchsim import AIMNet2TorchSim

        return AIMNet2TorchSim
    if name == "HipEngine":
        from .engine import HipEngine

        return HipEngine
    if name == "load_model":
        from .loader import load_model

        return load_model
    raise AttributeError(name)
