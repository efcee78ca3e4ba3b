"""Host-side mirror of the reference's Rust `gpu_manager` (src/gpu_manager/wrapper.rs, unit.rs, common.rs).

Same names, argument meaning and staging protocol as the Rust layer, over the same C ABI:
`PandaGpuManager` owns the device id, the memory pool and the four streams
(wrapper.rs:8-19, 31-53); `panda_msm_bn254_gpu*` / `panda_ntt_bn254_gpu*` stage byte slices to the
device, fill the by-value configuration struct, call the library and copy the answer back
(unit.rs:10-101, 103-188, 190-361, 363-416, 418-543).  Byte slices are numpy uint8/uint32 arrays.

Differences, all deliberate:
  * errors raise PandaGpuError instead of returning Err / unwrapping;
  * the pinned result buffer is freed (the reference leaks it, unit.rs:67-100);
  * cached scalars survive reuse, because the library no longer de-Montgomeryizes in place.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import gpu_ffi as ffi
from .gpu_ffi import JACOBIAN, PROJECTIVE, PandaGpuError  # noqa: F401

FIELD_ELEMENT_LEN = 32  # gpu_manager/mod.rs:14
BN254, BLS12_377, BLS12_381, BN254_G2 = 0, 1, 2, 3
_POINT_BYTES = {BN254: 64, BLS12_377: 96, BLS12_381: 96, BN254_G2: 128}
_RESULT_BYTES = {BN254: 96, BLS12_377: 144, BLS12_381: 144, BN254_G2: 192}


def _msm_entry(lib, curve: int, host: bool = False):
    names = {BN254: "bn254", BLS12_377: "bls12_377", BLS12_381: "bls12_381", BN254_G2: "bn254_g2"}
    return getattr(lib, f"panda_msm_execute_{names[curve]}" + ("_host" if host else ""))


def log_2(num: int) -> int:  # gpu_manager/common.rs:5-15
    assert num > 0
    return num.bit_length() - 1


def _as_bytes(a) -> np.ndarray:
    a = np.ascontiguousarray(a)
    return a.view(np.uint8).reshape(-1)


def _ptr(a: np.ndarray) -> C.c_void_p:
    return C.c_void_p(a.ctypes.data)


class PandaStreamHandle:
    """PandaStream::new/sync/wait/destroy (gpu_ffi/common.rs:46-87)."""

    def __init__(self):
        self.raw = ffi.PandaStream()
        ffi.check(ffi.load().panda_stream_create(C.byref(self.raw), True), "StremCreateErr")

    def sync(self):
        ffi.check(ffi.load().panda_stream_synchronize(self.raw), "StreamSyncErr")

    def wait(self, event: "PandaEventHandle"):
        ffi.check(ffi.load().panda_stream_wait_event(self.raw, event.raw), "StreamWaitEventErr")

    def destroy(self):
        ffi.check(ffi.load().panda_stream_destroy(self.raw), "StreamDestroyErr")


class PandaEventHandle:
    """PandaEvent::new/record/sync/destroy (gpu_ffi/common.rs:95-132)."""

    def __init__(self):
        self.raw = ffi.PandaEvent()
        ffi.check(ffi.load().panda_event_create(C.byref(self.raw), True, True), "EventCreateErr")

    def record(self, stream: PandaStreamHandle):
        ffi.check(ffi.load().panda_event_record(self.raw, stream.raw), "EventRecordErr")

    def sync(self):
        ffi.check(ffi.load().panda_event_sync(self.raw), "EventSyncErr")

    def destroy(self):
        ffi.check(ffi.load().panda_event_destroy(self.raw), "EventDestroyErr")


def get_device_number() -> int:  # wrapper.rs:315-323
    n = C.c_int(0)
    ffi.check(ffi.load().panda_get_device_number(C.byref(n)), "GetDeviceCountError")
    return n.value


def set_device(device_id: int) -> None:  # wrapper.rs:340-347
    ffi.check(ffi.load().panda_set_device(device_id), "SetDeviceError")


def device_info(device_id: int) -> dict:  # wrapper.rs:325-338
    set_device(device_id)
    free, total = C.c_size_t(0), C.c_size_t(0)
    ffi.check(ffi.load().panda_mem_get_info(C.byref(free), C.byref(total)), "DeviceGetDeviceMemoryInfoError")
    return {"free": free.value, "total": total.value}


class PandaGpuManager:
    def __init__(self, device_id: int = 0):  # PandaGpuManager::new, wrapper.rs:32-53
        if get_device_number() == 0:
            raise PandaGpuError("GetDeviceCountError")
        self.device_id = device_id
        self.mem_pool = self.init_hardware(device_id)
        self.default_stream = PandaStreamHandle()
        self.h2d_stream = PandaStreamHandle()
        self.d2h_stream = PandaStreamHandle()
        self.exec_stream = PandaStreamHandle()
        self.d_bases: list[int] = []
        self.d_scalars: list[int] = []
        self.scalars_len: list[int] = []
        self.msm_result_coordinate_type = JACOBIAN
        self._bases_bytes: dict[int, int] = {}
        self._registered: list[int] = []

    def add_cached_bases(self, bases) -> int:
        """init_msm_cached_bases + push onto d_bases (what callers of the reference do by hand); returns the index."""
        self.d_bases.append(self.init_msm_cached_bases(bases))
        self._bases_bytes[len(self.d_bases) - 1] = _as_bytes(bases).size
        return len(self.d_bases) - 1

    @classmethod
    def init_all(cls, device_id, bases=None, omega=None):  # wrapper.rs:55-113
        gm = cls(device_id)
        if bases is not None:
            gm.d_bases = cls.init_msm(bases)
        if omega is not None:
            cls.init_ntt(omega)
        if bases is None and omega is None:
            raise PandaGpuError("MSMBasesAddrError")
        return gm

    @staticmethod
    def init_hardware(device_id: int) -> ffi.PandaMemPool:  # wrapper.rs:115-120
        set_device(device_id)
        pool = ffi.PandaMemPool()
        ffi.check(ffi.load().panda_mem_pool_create(C.byref(pool), device_id), "MemPoolCreateErr")
        return pool

    @staticmethod
    def init_msm_cached_bases(bases) -> int:  # wrapper.rs:154-169
        b = _as_bytes(bases)
        d = C.c_void_p()
        lib = ffi.load()
        ffi.check(lib.panda_malloc(C.byref(d), b.size), "CreateContextError")
        ffi.check(lib.panda_memcpy(d, _ptr(b), b.size), "CreateContextError")
        return d.value

    init_msm_cached_scalars = init_msm_cached_bases  # wrapper.rs:171-186: same staging

    def register_cached_bases(self, index: int, curve: int = 0) -> None:
        """Additive: let the library keep the radix-converted copy of cached base set `index` between calls
        (panda_msm_register_bases).  The buffer must stay unmodified until deinit()."""
        d = self.get_params_bases_ptr_mut(index)
        if d is None:
            raise PandaGpuError("BasesIndexErr")
        nbytes = self._bases_bytes[index]
        log_n = log_2(nbytes // _POINT_BYTES[curve])
        ffi.check(ffi.load().panda_msm_register_bases(curve, C.c_void_p(d), log_n, self.exec_stream.raw), "CreateContextError")
        self._registered.append(d)

    def precompute_cached_bases(self, index: int, curve: int = 0, window_bits: int = 0) -> tuple[int, int, int]:
        """Additive: register cached base set `index` together with its window tables (panda_msm_precompute_bases).
        Returns (tables, window_bits, device bytes held)."""
        d = self.get_params_bases_ptr_mut(index)
        if d is None:
            raise PandaGpuError("BasesIndexErr")
        nbytes = self._bases_bytes[index]
        log_n = log_2(nbytes // _POINT_BYTES[curve])
        lib = ffi.load()
        ffi.check(lib.panda_msm_precompute_bases(curve, C.c_void_p(d), log_n, window_bits, self.exec_stream.raw), "CreateContextError")
        self._registered.append(d)
        tables, bits, held = C.c_uint(0), C.c_uint(0), C.c_size_t(0)
        ffi.check(lib.panda_msm_registered_info(C.c_void_p(d), C.byref(tables), C.byref(bits), C.byref(held)), "CreateContextError")
        return tables.value, bits.value, held.value

    @classmethod
    def init_msm(cls, bases_list) -> list[int]:  # wrapper.rs:122-152
        ptrs = [cls.init_msm_cached_bases(b) for b in bases_list]
        ffi.check(ffi.load().panda_msm_setup_bn254(), "CreateContextError")
        return ptrs

    @staticmethod
    def init_ntt(omega) -> None:  # wrapper.rs:199-210
        o = _as_bytes(omega)
        ffi.check(ffi.load().panda_ntt_setup_bn254(_ptr(o)), "CreateContextError")

    def set_config(self, coordinate_type: int):  # wrapper.rs:212-214
        self.msm_result_coordinate_type = coordinate_type

    def get_params_bases_ptr_mut(self, index: int):  # wrapper.rs:240-245
        return self.d_bases[index] if 0 <= index < len(self.d_bases) else None

    def get_params_scalars_ptr_mut(self, index: int):
        return self.d_scalars[index] if 0 <= index < len(self.d_scalars) else None

    def get_params_scalars_len(self, index: int) -> int:
        return self.scalars_len[index] if 0 <= index < len(self.scalars_len) else 0

    def wait_h2d(self):  # wrapper.rs:260-266
        ev = PandaEventHandle()
        ev.record(self.h2d_stream)
        self.exec_stream.wait(ev)
        ev.destroy()

    def sync(self):  # wrapper.rs:285-291
        self.h2d_stream.sync()
        self.exec_stream.sync()
        self.d2h_stream.sync()

    def deinit(self):  # wrapper.rs:297-312
        lib = ffi.load()
        for d in self._registered:
            lib.panda_msm_unregister_bases(C.c_void_p(d))  # "not registered" (the caller already undid it) is not an error here
        self._registered = []
        for d in self.d_bases + self.d_scalars:
            ffi.check(lib.panda_free(C.c_void_p(d)), "DestroyContextErr")
        self.d_bases, self.d_scalars, self.scalars_len = [], [], []
        ffi.check(lib.panda_msm_tear_down(), "DestroyContextErr")
        ffi.check(lib.panda_mem_pool_destroy(self.mem_pool), "DestroyContextErr")
        for s in (self.default_stream, self.h2d_stream, self.d2h_stream, self.exec_stream):
            s.destroy()


# ---------------------------------------------------------------------------- gpu_manager/common.rs

def memory_alloc_and_copy(gm: PandaGpuManager, h_values, stream: PandaStreamHandle) -> int:  # common.rs:54-65
    b = _as_bytes(h_values)
    d = C.c_void_p()
    lib = ffi.load()
    ffi.check(lib.panda_malloc_from_pool_async(C.byref(d), b.size, gm.mem_pool, stream.raw), "AsyncPoolMallocErr")
    ffi.check(lib.panda_memcpy_async(d, _ptr(b), b.size, stream.raw), "AsyncMemcopyErr")
    return d.value


def _pool_alloc(gm: PandaGpuManager, size: int, stream: PandaStreamHandle) -> int:
    d = C.c_void_p()
    ffi.check(ffi.load().panda_malloc_from_pool_async(C.byref(d), size, gm.mem_pool, stream.raw), "AsyncPoolMallocErr")
    return d.value


# ---------------------------------------------------------------------------- gpu_manager/unit.rs

def _msm_device(gm, d_scalars, d_bases, log_n, curve, free_scalars, free_bases):
    lib = ffi.load()
    nres = _RESULT_BYTES[curve]
    d_result = _pool_alloc(gm, nres, gm.h2d_stream)
    gm.wait_h2d()
    cfg = ffi.MSMConfiguration(gm.mem_pool, gm.exec_stream.raw, d_bases, d_scalars, d_result, log_n, gm.msm_result_coordinate_type)
    fn = _msm_entry(lib, curve)
    ffi.check(fn(cfg), "SchedulingErr")
    ev = PandaEventHandle()
    ev.record(gm.exec_stream)
    ev.sync()
    ev.destroy()
    host = C.c_void_p()
    ffi.check(lib.panda_malloc_host(C.byref(host), nres), "CreateContextError")
    try:
        ffi.check(lib.panda_memcpy(host, C.c_void_p(d_result), nres), "CreateContextError")
        out = np.frombuffer((C.c_uint8 * nres).from_address(host.value), dtype=np.uint8).copy()
    finally:
        lib.panda_free_host(host)
    for ptr, do in ((d_scalars, free_scalars), (d_bases, free_bases), (d_result, True)):
        if do:
            ffi.check(lib.panda_free(C.c_void_p(ptr)), "CreateContextError")
    return out


def panda_msm_bn254_gpu(gm: PandaGpuManager, scalars, bases, curve: int = BN254) -> np.ndarray:
    """unit.rs:10-101: stage scalars and bases, run, return the 96-byte X||Y||Z."""
    s = _as_bytes(scalars)
    d_scalars = memory_alloc_and_copy(gm, s, gm.h2d_stream)
    gm.wait_h2d()
    d_bases = memory_alloc_and_copy(gm, bases, gm.h2d_stream)
    gm.wait_h2d()
    return _msm_device(gm, d_scalars, d_bases, log_2(s.size // FIELD_ELEMENT_LEN), curve, True, True)


def pipeline_chunks(log_n: int) -> int:
    """Point ranges a single call's upload / execute pipeline is cut into: n/2^(R-1), n/2^(R-1), n/2^(R-2), ..., n/2 points (the
    library lowers R until the first range holds 2^16 points)."""
    return 5 if log_n >= 24 else (4 if log_n >= 22 else (3 if log_n >= 20 else (2 if log_n >= 18 else 1)))


def panda_msm_bn254_gpu_with_cached_bases(gm: PandaGpuManager, scalars, bases_index: int, curve: int = BN254) -> np.ndarray:
    """unit.rs:103-188.  When the cached base set is registered with the library (register_cached_bases /
    precompute_cached_bases) the upload and the execution are pipelined inside the one call: the scalars cross
    PCIe range by range on the h2d stream while the previous range is already being accumulated on the exec stream
    (panda_msm_execute_from_host); otherwise the reference's order -- upload everything, then execute."""
    s = _as_bytes(scalars)
    d_bases = gm.get_params_bases_ptr_mut(bases_index)
    if d_bases is None:
        raise PandaGpuError("BasesIndexErr")
    log_n = log_2(s.size // FIELD_ELEMENT_LEN)
    chunks = pipeline_chunks(log_n)
    if d_bases in gm._registered and chunks > 1:
        lib = ffi.load()
        nres = _RESULT_BYTES[curve]
        d_scalars = d_result = None
        try:  # both buffers go back on every way out, including a failed second allocation or wait
            d_scalars = _pool_alloc(gm, (1 << log_n) * FIELD_ELEMENT_LEN, gm.h2d_stream)
            d_result = _pool_alloc(gm, nres, gm.h2d_stream)
            gm.wait_h2d()
            cfg = ffi.MSMConfiguration(gm.mem_pool, gm.exec_stream.raw, d_bases, d_scalars, d_result, log_n, gm.msm_result_coordinate_type)
            ffi.check(lib.panda_msm_execute_from_host(curve, cfg, _ptr(s), chunks, gm.h2d_stream.raw), "SchedulingErr")
            out = np.zeros(nres, dtype=np.uint8)
            ffi.check(lib.panda_memcpy(_ptr(out), C.c_void_p(d_result), nres), "CreateContextError")
        finally:
            for d in (d_scalars, d_result):
                if d is not None:
                    lib.panda_free(C.c_void_p(d))
        return out
    d_scalars = memory_alloc_and_copy(gm, s, gm.h2d_stream)
    gm.wait_h2d()
    return _msm_device(gm, d_scalars, d_bases, log_n, curve, True, False)


def panda_msm_bn254_gpu_with_cached_bases_batched(gm: PandaGpuManager, scalars_batches, bases_index: int, curve: int = BN254) -> list:
    """Additive: a run of MSMs over one cached base set, using the manager's three streams the way
    wrapper.rs:12-14 intends them: the scalars of batch k+1 cross PCIe on the h2d stream while batch k executes on the
    exec stream; an event orders each execute after its own upload.  Two device buffers and one pinned staging buffer
    per slot, allocated once.  Returns the results in order; each equals panda_msm_bn254_gpu_with_cached_bases."""
    batches = [_as_bytes(b) for b in scalars_batches]
    if not batches:
        return []
    d_bases = gm.get_params_bases_ptr_mut(bases_index)
    if d_bases is None:
        raise PandaGpuError("BasesIndexErr")
    size = batches[0].size
    if any(b.size != size for b in batches):
        raise PandaGpuError("SchedulingErr")
    log_n = log_2(size // FIELD_ELEMENT_LEN)
    lib = ffi.load()
    nres = _RESULT_BYTES[curve]
    fn = _msm_entry(lib, curve)
    slots, results = [], []
    try:
        for _ in range(min(2, len(batches))):
            dev, pin, res = C.c_void_p(), C.c_void_p(), C.c_void_p()
            ffi.check(lib.panda_malloc(C.byref(dev), size), "AsyncPoolMallocErr")
            slots.append([dev, pin, res, PandaEventHandle()])
            ffi.check(lib.panda_malloc_host(C.byref(pin), size), "CreateContextError")
            slots[-1][1] = pin
            ffi.check(lib.panda_malloc_host(C.byref(res), nres), "CreateContextError")
            slots[-1][2] = res

        def upload(k):
            dev, pin, _, ev = slots[k & 1]
            C.memmove(pin, _ptr(batches[k]), size)  # pageable -> pinned, on the host while the GPU works
            ffi.check(lib.panda_memcpy_async(dev, pin, size, gm.h2d_stream.raw), "AsyncMemcopyErr")
            ev.record(gm.h2d_stream)

        upload(0)
        for k in range(len(batches)):
            dev, _, res, ev = slots[k & 1]
            ffi.check(lib.panda_stream_wait_event(gm.exec_stream.raw, ev.raw), "StreamWaitEventErr")
            if k + 1 < len(batches):
                upload(k + 1)  # the other slot: its previous execute has returned, so its buffers are free
            cfg = ffi.MSMConfiguration(gm.mem_pool, gm.exec_stream.raw, d_bases, dev, res, log_n, gm.msm_result_coordinate_type)
            ffi.check(fn(cfg), "SchedulingErr")  # synchronous: the result is in `res` on return
            results.append(np.frombuffer((C.c_uint8 * nres).from_address(res.value), dtype=np.uint8).copy())
    finally:
        gm.h2d_stream.sync()
        for dev, pin, res, ev in slots:
            ev.destroy()
            if dev:
                lib.panda_free(dev)
            if pin:
                lib.panda_free_host(pin)
            if res:
                lib.panda_free_host(res)
    return results


def panda_msm_bn254_gpu_with_cached_scalars(gm: PandaGpuManager, scalars_index: int, bases, curve: int = BN254) -> np.ndarray:
    """unit.rs:190-275: n comes from the bases length (len / 64, unit.rs:203)."""
    b = _as_bytes(bases)
    d_bases = memory_alloc_and_copy(gm, b, gm.h2d_stream)
    gm.wait_h2d()
    d_scalars = gm.get_params_scalars_ptr_mut(scalars_index)
    if d_scalars is None:
        raise PandaGpuError("BasesIndexErr")
    return _msm_device(gm, d_scalars, d_bases, log_2(b.size // _POINT_BYTES[curve]), curve, False, True)


def panda_msm_bn254_gpu_with_cached_input(gm: PandaGpuManager, scalars_index: int, bases_index: int, curve: int = BN254) -> np.ndarray:
    """unit.rs:277-361."""
    d_scalars = gm.get_params_scalars_ptr_mut(scalars_index)
    d_bases = gm.get_params_bases_ptr_mut(bases_index)
    if d_scalars is None or d_bases is None:
        raise PandaGpuError("BasesIndexErr")
    log_n = log_2(gm.get_params_scalars_len(scalars_index) // FIELD_ELEMENT_LEN)
    return _msm_device(gm, d_scalars, d_bases, log_n, curve, False, False)


def panda_msm_bn254_gpu_host(gm, scalars, bases, curve: int = BN254) -> np.ndarray:
    """unit.rs:363-416: the CPU host-debug entry point; every pointer is a host pointer.  `gm` may be None
    (the reference needs a live device only because of init_all, wrapper.rs:61-66)."""
    s, b = _as_bytes(scalars), _as_bytes(bases)
    lib = ffi.load()
    out = np.zeros(_RESULT_BYTES[curve], dtype=np.uint8)
    coord = gm.msm_result_coordinate_type if gm is not None else JACOBIAN
    cfg = ffi.MSMConfiguration(ffi.PandaMemPool(), ffi.PandaStream(), _ptr(b), _ptr(s), _ptr(out), log_2(s.size // FIELD_ELEMENT_LEN), coord)
    fn = _msm_entry(lib, curve, host=True)
    ffi.check(fn(cfg), "SchedulingErr")
    return out


def _ntt(gm, scalars: np.ndarray, log_n: int, call, omega=None):
    lib = ffi.load()
    buf = _as_bytes(scalars)
    assert buf.size == (1 << log_n) * 32  # unit.rs:423
    d_src = memory_alloc_and_copy(gm, buf, gm.h2d_stream)
    d_dst = _pool_alloc(gm, buf.size, gm.h2d_stream)
    flag = C.c_uint(0)
    if omega is None:
        cfg = ffi.NTTConfiguration(gm.mem_pool, gm.exec_stream.raw, d_src, d_dst, log_n, C.pointer(flag))
    else:
        o = _as_bytes(omega)
        cfg = ffi.NttconfigurationV1(gm.mem_pool, gm.exec_stream.raw, d_src, d_dst, _ptr(o), log_n, C.pointer(flag))
    ffi.check(call(cfg), "SchedulingErr")
    src = d_src if flag.value == 0 else d_dst  # unit.rs:521-532
    ffi.check(lib.panda_memcpy(_ptr(buf), C.c_void_p(src), buf.size), "CreateContextError")
    ffi.check(lib.panda_free(C.c_void_p(d_src)), "CreateContextError")
    ffi.check(lib.panda_free(C.c_void_p(d_dst)), "CreateContextError")
    return flag.value


def panda_ntt_bn254_gpu(gm: PandaGpuManager, scalars: np.ndarray, log_n: int) -> int:
    """unit.rs:418-479: in place on the caller's buffer, omega from init_ntt; returns the flag for inspection."""
    return _ntt(gm, scalars, log_n, ffi.load().panda_ntt_execute_bn254)


def panda_ntt_bn254_gpu_v1(gm: PandaGpuManager, scalars: np.ndarray, omega, log_n: int) -> int:
    """unit.rs:481-543."""
    return _ntt(gm, scalars, log_n, ffi.load().panda_ntt_execute_bn254_v1, omega)


def panda_coset_ntt_bn254_gpu(gm: PandaGpuManager, scalars: np.ndarray, omega, shift, log_n: int, inverse: bool = False) -> int:
    """Additive: coset transform, in place.  Forward y[k] = sum_j x[j] g^j w^(jk); inverse undoes it (n^-1 and g^-j fused)."""
    lib = ffi.load()
    g = _as_bytes(shift)
    fn = lib.panda_ntt_execute_bn254_coset_inverse if inverse else lib.panda_ntt_execute_bn254_coset
    return _ntt(gm, scalars, log_n, lambda cfg: fn(cfg, _ptr(g)), omega)


def panda_ntt_bls12_381_gpu_v1(gm: PandaGpuManager, scalars: np.ndarray, omega, log_n: int, inverse: bool = False) -> int:
    """Additive: the v1 transform over the BLS12-381 scalar field (two-adicity 32)."""
    lib = ffi.load()
    return _ntt(gm, scalars, log_n, lib.panda_ntt_execute_bls12_381_inverse if inverse else lib.panda_ntt_execute_bls12_381_v1, omega)


def panda_ntt_bls12_377_gpu_v1(gm: PandaGpuManager, scalars: np.ndarray, omega, log_n: int, inverse: bool = False) -> int:
    """Additive: the same staging for the BLS12-377 scalar field."""
    lib = ffi.load()
    return _ntt(gm, scalars, log_n, lib.panda_ntt_execute_bls12_377_inverse if inverse else lib.panda_ntt_execute_bls12_377_v1, omega)


def panda_ntt_bls12_377_gpu_bitrev(gm: PandaGpuManager, scalars: np.ndarray, omega, log_n: int, inverse: bool = False, field: str = "bls12_377") -> int:
    """Additive: the bit-reversed orderings of panda_ntt_bn254_gpu_bitrev over the BLS12-377 (or, field="bls12_381", BLS12-381) scalar field."""
    lib = ffi.load()
    return _ntt(gm, scalars, log_n, getattr(lib, f"panda_ntt_execute_{field}_inverse_bitrev_in" if inverse else f"panda_ntt_execute_{field}_bitrev_out"), omega)


def panda_coset_ntt_bls12_377_gpu(gm: PandaGpuManager, scalars: np.ndarray, omega, shift, log_n: int, inverse: bool = False, field: str = "bls12_377") -> int:
    """Additive: coset transform over the BLS12-377 (or BLS12-381) scalar field (as panda_coset_ntt_bn254_gpu)."""
    lib = ffi.load()
    g = _as_bytes(shift)
    fn = getattr(lib, f"panda_ntt_execute_{field}_coset_inverse" if inverse else f"panda_ntt_execute_{field}_coset")
    return _ntt(gm, scalars, log_n, lambda cfg: fn(cfg, _ptr(g)), omega)


def panda_intt_bn254_gpu(gm: PandaGpuManager, scalars: np.ndarray, omega, log_n: int) -> int:
    """Additive: inverse transform with the n^-1 scaling fused (panda_ntt_execute_bn254_inverse)."""
    return _ntt(gm, scalars, log_n, ffi.load().panda_ntt_execute_bn254_inverse, omega)


def panda_ntt_bn254_gpu_bitrev(gm: PandaGpuManager, scalars: np.ndarray, omega, log_n: int, inverse: bool = False) -> int:
    """Additive: forward transform leaving y[k] at bitrev(k), or (inverse=True) the inverse of a buffer in that order back to
    natural-order coefficients with n^-1 fused.  In place on the caller's buffer like panda_ntt_bn254_gpu_v1."""
    lib = ffi.load()
    return _ntt(gm, scalars, log_n, lib.panda_ntt_execute_bn254_inverse_bitrev_in if inverse else lib.panda_ntt_execute_bn254_bitrev_out, omega)
