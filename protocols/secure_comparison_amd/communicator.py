"""Transport contract between the two players (reference: communicator.py:8-11).  Out of scope as a product
component -- any object with these two coroutines works (the tests use an in-memory dictionary)."""
from __future__ import annotations

from typing import Any, Protocol


class Communicator(Protocol):
    async def send(self, party_id: str, message: Any, msg_id: str | None = None) -> None: ...

    async def recv(self, party_id: str, msg_id: str | None = None) -> Any: ...
