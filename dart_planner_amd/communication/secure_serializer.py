"""Wire format of the cloud <-> edge link: the reference's HMAC-signed JSON envelope
(src/dart_planner/communication/secure_serializer.py:29-194), byte-compatible in both directions, plus the
Trajectory <-> dict mapping the reference lacks.

Envelope (reference :92-129): JSON object {"data", "signature", "timestamp", "message_id"} where
signature = HMAC-SHA256(secret, f"{json.dumps(data)}:{timestamp}:{message_id}").hexdigest(); the receiver
re-serialises the parsed ``data`` with ``json.dumps`` and compares (:146-164), rejects messages older than
the TTL (300 s default, env DART_MSG_TTL), and turns flat lists of numbers back into ndarrays (:205-222).

The reference's cloud handler returns ``{"trajectory": <Trajectory dataclass>}`` (cloud/main_improved.py:87),
which its own serializer cannot encode (``_json_serializer`` raises TypeError for dataclasses, :169-177).
``trajectory_to_wire`` / ``trajectory_from_wire`` define the missing mapping: one key per Trajectory field,
arrays as nested lists, absent fields as null.  Host-side Python; ZeroMQ itself is not part of this package.
"""
import hashlib
import hmac
import json
import os
import time
from dataclasses import asdict, dataclass, fields
from typing import Any, Optional

import numpy as np

from ..common.errors import DARTPlannerError
from ..common.types import Trajectory


class CommunicationError(DARTPlannerError):
    pass


class SecurityError(DARTPlannerError):
    pass


@dataclass
class SecureMessage:
    data: Any
    signature: str
    timestamp: float
    message_id: str


class SecureSerializer:
    def __init__(self, secret_key: Optional[str] = None, test_mode: bool = False, message_ttl: Optional[int] = None):
        env_secret = os.getenv("DART_ZMQ_SECRET")
        self._test_mode = test_mode or os.getenv("DART_ENVIRONMENT", "development") in ("test", "testing")
        if secret_key:
            self.secret_key = secret_key
        elif env_secret:
            self.secret_key = env_secret
        elif not self._test_mode:
            raise SecurityError("DART_ZMQ_SECRET must be set in non-test environments for secure ZMQ communication.")
        else:
            import secrets
            self.secret_key = secrets.token_urlsafe(32)
        self._message_counter = 0
        if message_ttl is not None:
            self._msg_ttl = message_ttl
        else:
            try:
                self._msg_ttl = int(os.getenv("DART_MSG_TTL") or 300)
            except ValueError:
                self._msg_ttl = 300

    # reference :72-90
    def _generate_message_id(self) -> str:
        self._message_counter += 1
        return f"msg_{self._message_counter}_{os.getpid()}"

    def _sign_data(self, data: str, timestamp: float, message_id: str) -> str:
        return hmac.new(self.secret_key.encode("utf-8"), f"{data}:{timestamp}:{message_id}".encode("utf-8"),
                        hashlib.sha256).hexdigest()

    def _verify_signature(self, data: str, timestamp: float, message_id: str, signature: str) -> bool:
        return hmac.compare_digest(signature, self._sign_data(data, timestamp, message_id))

    # reference :92-129
    def serialize(self, obj: Any, *, timestamp: Optional[float] = None, message_id: Optional[str] = None) -> bytes:
        obj = _plain(obj)
        timestamp = time.time() if timestamp is None else timestamp
        message_id = self._generate_message_id() if message_id is None else message_id
        data_json = json.dumps(obj, default=_json_default)
        msg = SecureMessage(data=obj, signature=self._sign_data(data_json, timestamp, message_id), timestamp=timestamp,
                            message_id=message_id)
        return json.dumps(asdict(msg), default=_json_default).encode("utf-8")

    # reference :131-167
    def deserialize(self, data: bytes, *, now: Optional[float] = None) -> Any:
        try:
            msg = SecureMessage(**json.loads(data.decode("utf-8")))
        except (json.JSONDecodeError, TypeError, UnicodeDecodeError) as e:
            raise CommunicationError(f"Invalid message format: {e}")
        if (time.time() if now is None else now) - msg.timestamp > self._msg_ttl:
            raise CommunicationError("Message too old")
        if not self._verify_signature(json.dumps(msg.data, default=_json_default), msg.timestamp, msg.message_id, msg.signature):
            raise CommunicationError("Message signature verification failed")
        return _restore(msg.data)


def _json_default(obj):
    """reference :169-177"""
    if isinstance(obj, np.ndarray):
        return obj.tolist()
    if isinstance(obj, np.integer):
        return int(obj)
    if isinstance(obj, np.floating):
        return float(obj)
    raise TypeError(f"Object of type {type(obj)} is not JSON serializable")


def _plain(obj):
    """ndarrays -> lists, recursively through dicts and lists (reference :179-203); a Trajectory is mapped to
    its wire dict (the extension that makes the cloud handler's reply encodable)."""
    if isinstance(obj, Trajectory):
        return trajectory_to_wire(obj)
    if isinstance(obj, np.ndarray):
        return obj.tolist()
    if isinstance(obj, dict):
        return {k: _plain(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_plain(v) for v in obj]
    return obj


def _restore(obj, depth: int = 0):
    """reference :205-222: a flat list of numbers becomes an ndarray; bounded recursion."""
    if depth > 100:
        raise CommunicationError("Maximum recursion depth 100 exceeded during deserialization")
    if isinstance(obj, list):
        if all(isinstance(x, (int, float)) for x in obj):
            return np.array(obj)
        return [_restore(x, depth + 1) for x in obj]
    if isinstance(obj, dict):
        return {k: _restore(v, depth + 1) for k, v in obj.items()}
    return obj


_TRAJ_FIELDS = [f.name for f in fields(Trajectory)]


def trajectory_to_wire(tr: Trajectory) -> dict:
    return {n: (None if getattr(tr, n) is None else np.asarray(getattr(tr, n), float).tolist()) for n in _TRAJ_FIELDS}


def trajectory_from_wire(d: dict) -> Trajectory:
    def arr(v):
        if v is None:
            return None
        return np.asarray([np.asarray(r, float) for r in v] if isinstance(v, list) and v and isinstance(v[0], (list, np.ndarray)) else v, float)
    missing = [n for n in ("timestamps", "positions") if d.get(n) is None]
    if missing:
        raise CommunicationError(f"trajectory message lacks {missing}")
    return Trajectory(**{n: arr(d.get(n)) for n in _TRAJ_FIELDS})
