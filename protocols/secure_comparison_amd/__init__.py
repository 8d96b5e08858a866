"""MI355X-native batched secure comparison (DGK/Veugen protocol): the Paillier / DGK arithmetic underneath
Initiator.step_* and KeyHolder.step_* as hand-written HIP kernels behind a C ABI (libsc_amd.so)."""
from .communicator import Communicator, InMemoryCommunicator, StreamCommunicator
from .initiator import AlicePlain, Initiator
from .keyholder import BobPlain, KeyHolder
from .schemes import DGK, DGKCiphertext, Paillier, PaillierCiphertext
from .utils import from_bits, to_bits

__all__ = ["Communicator", "InMemoryCommunicator", "StreamCommunicator", "Initiator", "KeyHolder", "from_bits", "to_bits", "Paillier", "PaillierCiphertext", "DGK",
           "DGKCiphertext", "AlicePlain", "BobPlain"]
__version__ = "0.1.0"
