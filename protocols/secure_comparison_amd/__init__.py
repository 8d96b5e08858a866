"""MI355X-native batched secure comparison (DGK/Veugen protocol): the Paillier / DGK arithmetic underneath
Initiator.step_* and KeyHolder.step_* as hand-written HIP kernels behind a C ABI (libsc_amd.so)."""
import os as _os

# The HIP runtime multiplexes a process's streams onto 4 hardware queues by default, and streams that land on one queue run in
# sequence.  Two library contexts with their fork streams are four streams already (configs[1]: 114 k/s with a queue each, 80 k/s
# when a fifth stream of the same process made two of them share one); two byte-transport sessions with their copy streams are
# more; every context that generates randomizers in the background is another.  Read by the runtime when it initialises, i.e. at the first GPU call of the process; an exported value wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")

from .communicator import Communicator, InMemoryCommunicator, StreamCommunicator
from .initiator import AlicePlain, Initiator
from .keyholder import BobPlain, KeyHolder
from .schemes import DGK, DGKCiphertext, Paillier, PaillierCiphertext
from .utils import from_bits, to_bits

__all__ = ["Communicator", "InMemoryCommunicator", "StreamCommunicator", "Initiator", "KeyHolder", "from_bits", "to_bits", "Paillier", "PaillierCiphertext", "DGK",
           "DGKCiphertext", "AlicePlain", "BobPlain"]
__version__ = "0.1.0"
