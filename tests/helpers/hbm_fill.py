#!/usr/bin/env python3
"""Test helper: a FOREIGN process that takes device memory until only `keep_mb` MB are free, prints "ready <free MB>" and holds it until stdin closes.
usage: hbm_fill.py keep_mb [device]"""
import ctypes as C, sys
def main():
    keep = int(sys.argv[1]) << 20; dev = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]; hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    if hip.hipSetDevice(dev) != 0: print("error setdevice", flush=True); return 1
    held = []
    for _ in range(64):                      # in pieces: one request of a few hundred GB may be refused where its parts are not
        fr, tot = C.c_size_t(0), C.c_size_t(0)
        if hip.hipMemGetInfo(C.byref(fr), C.byref(tot)) != 0: print("error meminfo", flush=True); return 1
        if fr.value <= keep + (8 << 20): break
        want = min(fr.value - keep, 32 << 30)
        p = C.c_void_p(0)
        if hip.hipMalloc(C.byref(p), want) != 0:
            want //= 2
            if want < (16 << 20) or hip.hipMalloc(C.byref(p), want) != 0: break
        held.append(p)
    fr, tot = C.c_size_t(0), C.c_size_t(0); hip.hipMemGetInfo(C.byref(fr), C.byref(tot))
    print("ready %d" % (fr.value >> 20), flush=True)
    sys.stdin.read()
    return 0
if __name__ == "__main__": sys.exit(main())
