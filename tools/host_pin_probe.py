#!/usr/bin/env python3
"""Which GPU virtual address does page-locked HOST memory get on this stack?  hipHostRegister on an array in the glibc
heap and on an mmap'ed one; host pointer next to hipHostGetDevicePointer.  (Reading a GPU memory-fault address: is it
a host address?)     python tools/host_pin_probe.py"""
import ctypes as C
import mmap

import numpy as np
import torch

torch.cuda.init()
hip = C.CDLL(torch.__file__.rsplit("/", 1)[0] + "/lib/libamdhip64.so")
for name, arr in (("heap 200 KB", np.zeros(25_000)), ("heap 8 MB", np.zeros(1_000_000)),
                  ("mmap 8 MB", np.frombuffer(mmap.mmap(-1, 8_000_000), dtype=np.float64))):
    p = arr.ctypes.data
    e = hip.hipHostRegister(C.c_void_p(p), C.c_size_t(arr.nbytes), C.c_uint(0))
    d = C.c_void_p()
    e2 = hip.hipHostGetDevicePointer(C.byref(d), C.c_void_p(p), C.c_uint(0))
    print(f"{name}: host {p:#x} device {d.value or 0:#x} same={d.value == p} (register {e}, get {e2})", flush=True)
    hip.hipHostUnregister(C.c_void_p(p))
print(open("/proc/self/maps").read().split("[heap]")[0].splitlines()[-1], "[heap]")
