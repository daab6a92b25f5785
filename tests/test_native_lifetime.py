"""Order of Context.close() and the release of page-locked arrays (no GPU: the native library is played by a
recorder).  vw_host_free must never be called with a destroyed context, and vw_destroy must run exactly once,
after the last array obtained from host_array() is gone."""
import ctypes as C
import gc

from vectorwave_amd import _native as N


class _FakeLib:
    def __init__(self):
        self.calls = []
        self.live = True
        self._mem = []

    def vw_host_alloc(self, handle, nbytes, out):
        buf = (C.c_char * nbytes)()
        self._mem.append(buf)
        C.cast(out, C.POINTER(C.c_void_p))[0] = C.addressof(buf)
        self.calls.append(("alloc", nbytes))
        return 0

    def vw_host_free(self, handle, ptr):
        assert self.live, "vw_host_free on a destroyed context"
        self.calls.append(("free", ptr.value))
        return 0

    def vw_destroy(self, handle):
        assert self.live, "vw_destroy twice"
        self.live = False
        self.calls.append(("destroy",))
        return 0


def _context(lib):
    ctx = object.__new__(N.Context)
    ctx.lib = lib
    ctx.handle = C.c_void_p(1234)
    ctx.device = 0
    ctx._state = N._ContextState(lib, ctx.handle)
    return ctx


def test_close_before_array_release_defers_destroy():
    lib = _FakeLib()
    ctx = _context(lib)
    a = ctx.host_array((16,))
    b = ctx.host_array((4, 4))
    ctx.close()
    assert ctx.handle is None and lib.live, "the native context must outlive its page-locked arrays"
    del a
    gc.collect()
    assert lib.live
    del b
    gc.collect()
    assert not lib.live
    assert [c[0] for c in lib.calls] == ["alloc", "alloc", "free", "free", "destroy"]


def test_array_release_before_close_destroys_at_close():
    lib = _FakeLib()
    ctx = _context(lib)
    a = ctx.host_array((8,))
    del a
    gc.collect()
    assert lib.live and [c[0] for c in lib.calls] == ["alloc", "free"]
    ctx.close()
    assert not lib.live
    ctx.close()  # idempotent
    assert [c[0] for c in lib.calls] == ["alloc", "free", "destroy"]


def test_context_collected_with_live_array():
    lib = _FakeLib()
    ctx = _context(lib)
    a = ctx.host_array((8,))
    del ctx
    gc.collect()
    assert lib.live
    a[:] = 1.0  # the memory is still there
    del a
    gc.collect()
    assert not lib.live
