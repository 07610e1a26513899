"""TEST INFRASTRUCTURE ONLY: puts the CPU build of the kernel sources (build_emu.py) behind battgp_amd._lib for the
duration of a test.  Nothing in battgp_amd/ knows about this module; the product has no way to reach it."""

from __future__ import annotations

import ctypes as C
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

_cache: dict = {}


def load_emu(sanitize: bool = False, experimental=None) -> C.CDLL:
    import build_emu

    from battgp_amd import _lib

    if experimental is None:
        experimental = build_emu.experimental_default()
    key = (str(sanitize) if sanitize else "plain") + ("+exp" if experimental else "")
    if key not in _cache:
        lib = C.CDLL(build_emu.build(sanitize=sanitize, experimental=experimental))
        lib.hipemu_stream_sync.argtypes = [C.c_void_p]
        for name, (res, args) in _lib.SIGNATURES.items():
            fn = getattr(lib, name)  # the CPU build must export the whole C-ABI too
            fn.restype = res
            fn.argtypes = args
        _cache[key] = lib
    return _cache[key]


class installed:
    """``with installed(): ...`` - ExactGPEngine & co. talk to the CPU build inside the block.
    ``experimental=True``: the CPU build of the experimental library (the optional kernel families); None: as the environment
    says (BGP_EMU_EXPERIMENTAL=1), default = the product's configuration."""

    def __init__(self, sanitize: bool = False, experimental=None):
        self.sanitize, self.experimental = sanitize, experimental

    def __enter__(self):
        from battgp_amd import _lib

        self._saved = _lib._lib
        _lib._lib = load_emu(self.sanitize, self.experimental)
        return _lib._lib

    def __exit__(self, *exc):
        from battgp_amd import _lib

        _lib._lib = self._saved
        return False


def fake_cuda_tensors():
    """Development aid for ``pytest --emu``: the CPU build's "device memory" is host memory, so a test's
    ``torch.zeros(..., device="cuda")`` / ``.cuda()`` / ``.to(cuda_device)`` may simply stay on the host and
    ``data_ptr()`` is a valid "device pointer".  Patches the handful of torch entry points the tests use."""
    import contextlib

    import torch

    def is_cuda(dev):
        if dev is None:
            return False
        if isinstance(dev, int):
            return True
        try:
            return torch.device(dev).type == "cuda"
        except Exception:
            return False

    def strip(fn):
        def wrapped(*a, **k):
            if is_cuda(k.get("device")):
                k = dict(k, device="cpu")
            return fn(*a, **k)

        return wrapped

    for name in ("zeros", "empty", "full", "ones", "tensor", "as_tensor", "arange", "randn", "rand", "empty_like", "zeros_like", "linspace", "eye"):
        setattr(torch, name, strip(getattr(torch, name)))
    if os.environ.get("HIPEMU_POISON", "").lower() == "ff":
        # "device" tensors the drivers allocate uninitialised (torch.empty on the host: usually fresh zero pages) become NaN,
        # like the CPU build's hipMalloc under the same switch - up to 2^28 elements (the huge-stride checks stay virtual)
        plain_empty = torch.empty

        def poisoned_empty(*a, **k):
            t = plain_empty(*a, **k)
            if t.is_floating_point() and 0 < t.numel() <= (1 << 28):
                t.fill_(float("nan"))
            return t

        torch.empty = poisoned_empty
    real_to = torch.Tensor.to

    def to(self, *a, **k):
        a = tuple("cpu" if (not isinstance(x, (torch.dtype, torch.Tensor, bool)) and is_cuda(x) and not isinstance(x, int)) else x for x in a)
        if is_cuda(k.get("device")):
            k = dict(k, device="cpu")
        return real_to(self, *a, **k)

    torch.Tensor.to = to
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.cuda.is_available = lambda: True
    torch.cuda.device_count = lambda: 1
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.set_device = lambda *a, **k: None
    torch.cuda.current_device = lambda: 0
    torch.cuda.empty_cache = lambda: None
    torch.cuda.mem_get_info = lambda *a, **k: (48 << 30, 64 << 30)

    # torch work "on the engine's stream" (sharded.py makes the engine's stream current around its buffer operations
    # and collectives) is host work here: it runs behind everything the stream already holds
    @contextlib.contextmanager
    def on_stream(s):
        from battgp_amd import _lib

        ptr = getattr(s, "ptr", None)
        cur = _lib._lib  # the CPU build that is installed right now (default or experimental configuration)
        if ptr and cur is not None and hasattr(cur, "hipemu_stream_sync"):
            cur.hipemu_stream_sync(ptr)
        yield

    torch.cuda.stream = on_stream
    torch.cuda.is_current_stream_capturing = lambda: False  # torch.optim's capture health check
    torch.Tensor.is_cuda = property(lambda self: True)      # "device" tensors live on the host here

    class _Stream:
        def __init__(self, ptr=None, *a, **k):
            self.ptr = ptr

        def synchronize(self):
            pass

        def wait_stream(self, other):
            pass

    torch.cuda.ExternalStream = _Stream
    torch.cuda.current_stream = lambda *a, **k: _Stream()
