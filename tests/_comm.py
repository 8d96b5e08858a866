"""In-memory transport for the two players (test helper; the reference's tests use a similar dictionary fake,
test/conftest.py:162-198)."""
import asyncio


class DictionaryCommunicator:
    def __init__(self, box: dict):
        self.box = box

    async def send(self, party_id, message, msg_id=None):
        self.box[msg_id] = message

    async def recv(self, party_id, msg_id=None):
        for _ in range(100000):
            if msg_id in self.box:
                return self.box.pop(msg_id)
            await asyncio.sleep(0)
        raise TimeoutError(msg_id)
