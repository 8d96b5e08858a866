"""In-memory transport for the two players (thin alias; the reference's tests use a similar dictionary fake,
test/conftest.py:162-198).  device_tensors=False makes the batch protocol serialize its arrays to bytes."""
from protocols.secure_comparison_amd.communicator import InMemoryCommunicator


class DictionaryCommunicator(InMemoryCommunicator):
    def __init__(self, box: dict, device_tensors: bool = True):
        super().__init__(box, device_tensors=device_tensors)
