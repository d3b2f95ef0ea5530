"""The HIP virtual-memory API straight through ctypes, for the placement probes (vmm_offset_sweep.py,
vmm_region_probe.py, vmm_select_probe.py, vmm_greedy_probe.py): pools of 2 MiB physical handles and
windows mapped onto them.  Probes only -- the product's entry is fcamd_device_alloc_set.  Address ranges are
never reused (a re-reserved range serves stale data on this stack)."""
import ctypes as C

import torch

hip = C.CDLL("libamdhip64.so.7")  # the runtime torch has loaded
G = 2 << 20


class Loc(C.Structure):
    _fields_ = [("type", C.c_int), ("id", C.c_int)]


class Prop(C.Structure):
    _fields_ = [("type", C.c_int), ("handle_type", C.c_int), ("location", Loc), ("win32", C.c_void_p),
                ("compression", C.c_ubyte), ("rdma", C.c_ubyte), ("usage", C.c_ushort)]


class Access(C.Structure):
    _fields_ = [("location", Loc), ("flags", C.c_int)]


PROP, ACC = Prop(1, 0, Loc(1, 0), None, 0, 0, 0), Access(Loc(1, 0), 3)  # pinned on device 0; read-write


def chk(e, what):
    if e != 0:
        raise RuntimeError(f"{what} -> hip error {e}")


def create_handles(count):
    """`count` physical 2 MiB handles, created in one burst"""
    out = []
    for _ in range(count):
        h = C.c_void_p()
        chk(hip.hipMemCreate(C.byref(h), C.c_size_t(G), C.byref(PROP), C.c_ulonglong(0)), "hipMemCreate")
        out.append(h)
    return out


def map_handles(handles):
    """a fresh address range with page i backed by handles[i]; returns its base address"""
    va = C.c_void_p()
    chk(hip.hipMemAddressReserve(C.byref(va), C.c_size_t(len(handles) * G), C.c_size_t(G), None, C.c_ulonglong(0)), "reserve")
    for i, h in enumerate(handles):
        p = C.c_void_p(va.value + i * G)
        chk(hip.hipMemMap(p, C.c_size_t(G), C.c_size_t(0), h, C.c_ulonglong(0)), "map")
        chk(hip.hipMemSetAccess(p, C.c_size_t(G), C.byref(ACC), C.c_size_t(1)), "access")
    return va.value


def unmap(va, count, release=None):
    """unmap the `count` pages at va (after a device synchronise); `release`: handles to give back as well"""
    torch.cuda.synchronize()
    for i in range(count):
        chk(hip.hipMemUnmap(C.c_void_p(va + i * G), C.c_size_t(G)), "unmap")
    for h in release or []:
        chk(hip.hipMemRelease(h), "release")


def bandwidth_GBs(fn, nbytes, reps=4):
    fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return nbytes / min(a.elapsed_time(b) for a, b in ev) / 1e6
