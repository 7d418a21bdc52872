"""Host-buffer array backend for dart_planner_amd.ops.Ops -- TEST INFRASTRUCTURE.
Pairs with tests/emu/libse3mpc_emu.so (the product kernels compiled for the host)."""
import numpy as np


class NumpyBackend:
    _dt = {"f32": np.float32, "f64": np.float64, "i32": np.int32, "i64": np.int64, "u8": np.uint8}

    def empty(self, shape, kind):
        """Like torch.empty, but poisoned: every byte is 0x7B, so a kernel that forgets to write shows up."""
        nbytes = int(np.prod(shape)) * np.dtype(self._dt[kind]).itemsize
        return np.frombuffer(bytearray(b"\x7b" * nbytes), dtype=self._dt[kind]).reshape(shape)

    def suffix(self, a):
        if a.dtype == np.float32:
            return "f32"
        if a.dtype == np.float64:
            return "f64"
        raise TypeError(a.dtype)

    def check(self, a, name):
        if not isinstance(a, np.ndarray) or not a.flags["C_CONTIGUOUS"]:
            raise ValueError(f"{name}: need a C-contiguous ndarray")
        return a

    def ptr(self, a):
        return 0 if a is None else a.ctypes.data

    def is_host_mapped(self, a):
        return isinstance(a, np.ndarray) and a.flags["C_CONTIGUOUS"]      # the emulated device IS host memory

    def stream(self):
        return 0

    def to_host(self, a):
        return a

    def from_host(self, a):
        return np.ascontiguousarray(a).copy()


class TorchCpuBackend:
    """CPU torch tensors + the emulated library: lets the CPU suite drive the planner's real host
    code path (which speaks torch) without a GPU.  TEST INFRASTRUCTURE (the product's TorchBackend
    refuses CPU tensors)."""

    def __init__(self):
        import torch
        self.torch = torch
        self.device = torch.device("cpu")
        self._dt = {"f32": torch.float32, "f64": torch.float64, "i32": torch.int32, "i64": torch.int64, "u8": torch.uint8}

    def empty(self, shape, kind):
        return self.torch.empty(shape, dtype=self._dt[kind])

    def suffix(self, a):
        return "f32" if a.dtype == self.torch.float32 else "f64"

    def check(self, a, name):
        if not a.is_contiguous():
            raise ValueError(f"{name}: tensor must be contiguous")
        return a

    def ptr(self, a):
        return 0 if a is None else a.data_ptr()

    def is_host_mapped(self, a):
        return self.torch.is_tensor(a) and a.is_contiguous()      # the emulated device IS host memory

    def stream(self):
        return 0

    def to_host(self, a):
        return a.numpy()

    def elements_from(self, a):
        return (a.untyped_storage().nbytes() - a.storage_offset() * a.element_size()) // a.element_size()

    def from_host(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(a).copy())
