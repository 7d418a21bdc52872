"""CPU: the cloud<->edge wire format against messages produced by the reference's own
SecureSerializer (tests/golden/wire_messages.json): this package verifies and decodes them, and
re-serialising the decoded payload with the same (timestamp, message_id) reproduces the reference's
bytes, i.e. an unmodified edge client accepts what this package signs."""
import json
import os

import numpy as np
import pytest

from dart_planner_amd.common.types import Trajectory
from dart_planner_amd.communication.secure_serializer import (CommunicationError, SecureSerializer, SecurityError,
                                                              trajectory_from_wire, trajectory_to_wire)

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "wire_messages.json")


def test_reference_messages_verify_and_round_trip_bytewise():
    g = json.load(open(GOLDEN))
    ser = SecureSerializer(secret_key=g["secret"], message_ttl=10 ** 9)
    for m in g["messages"]:
        raw = m["raw"].encode("utf-8")
        env = json.loads(raw)
        obj = ser.deserialize(raw)
        again = ser.serialize(env["data"], timestamp=env["timestamp"], message_id=env["message_id"])
        assert again == raw
        assert json.loads(again)["signature"] == env["signature"]
        if isinstance(obj, dict) and "goal" in obj:
            assert isinstance(obj["goal"], np.ndarray) and np.array_equal(obj["goal"], [5.0, 3.0, 2.0])
            assert isinstance(obj["nested"]["b"][0], np.ndarray)          # nested lists: inner flat lists become arrays (:205-222)


def test_tamper_ttl_and_secret():
    ser = SecureSerializer(secret_key="k1", message_ttl=300)
    raw = ser.serialize({"x": np.array([1.0, 2.0])})
    assert np.array_equal(ser.deserialize(raw)["x"], [1.0, 2.0])
    env = json.loads(raw); env["data"]["x"][0] = 9.0
    with pytest.raises(CommunicationError, match="signature"):
        ser.deserialize(json.dumps(env).encode())
    with pytest.raises(CommunicationError, match="too old"):
        ser.deserialize(raw, now=json.loads(raw)["timestamp"] + 301)
    with pytest.raises(CommunicationError, match="signature"):
        SecureSerializer(secret_key="k2").deserialize(raw)
    with pytest.raises(CommunicationError, match="Invalid message"):
        ser.deserialize(b"not json")
    old = os.environ.pop("DART_ZMQ_SECRET", None), os.environ.pop("DART_ENVIRONMENT", None)
    try:
        with pytest.raises(SecurityError):
            SecureSerializer()
        assert SecureSerializer(test_mode=True).secret_key
    finally:
        for k, v in zip(("DART_ZMQ_SECRET", "DART_ENVIRONMENT"), old):
            if v is not None:
                os.environ[k] = v


def test_trajectory_round_trip_through_the_envelope():
    rng = np.random.default_rng(0)
    N = 6
    tr = Trajectory(timestamps=1.7e9 + np.arange(N) / 400, positions=rng.normal(size=(N, 3)), velocities=rng.normal(size=(N, 3)),
                    accelerations=rng.normal(size=(N, 3)), attitudes=rng.normal(size=(N, 3)), body_rates=None,
                    thrusts=rng.uniform(2, 25, N), yaws=rng.normal(size=N), yaw_rates=None)
    ser = SecureSerializer(secret_key="k")
    back = trajectory_from_wire(ser.deserialize(ser.serialize({"trajectory": tr}))["trajectory"])     # the cloud handler's reply shape
    for f in ("timestamps", "positions", "velocities", "accelerations", "attitudes", "thrusts", "yaws"):
        assert np.array_equal(getattr(back, f), getattr(tr, f)), f                                     # repr round trip: exact
    assert back.body_rates is None and back.yaw_rates is None
    assert set(trajectory_to_wire(tr)) == {"timestamps", "positions", "velocities", "accelerations", "attitudes", "body_rates",
                                           "thrusts", "yaws", "yaw_rates"}
    with pytest.raises(CommunicationError):
        trajectory_from_wire({"positions": None, "timestamps": [0.0]})
