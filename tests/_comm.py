"""In-memory transport for the two players (thin alias; the reference's tests use a similar dictionary fake,
test/conftest.py:162-198)."""
from protocols.secure_comparison_amd.communicator import InMemoryCommunicator


class DictionaryCommunicator(InMemoryCommunicator):
    def __init__(self, box: dict):
        super().__init__(box)
