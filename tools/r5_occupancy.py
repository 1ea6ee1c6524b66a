"""Resident workgroups per CU the launcher gets from the runtime for the Stockham / register-tiled routes, next to what 160 KiB of LDS, 2048
threads and the kernel's registers would admit (development tool)."""
import os, sys, re
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pffft_amd as pa
L = pa.lib()
L.pffft_hip_route_occupancy.restype = int
import ctypes as C
L.pffft_hip_route_occupancy.argtypes = [C.c_void_p, C.c_int, C.c_int]
torch.cuda.init(); torch.zeros(1, device="cuda")
for dt in (np.float32, np.float64):
    for tr in (pa.COMPLEX, pa.REAL):
        for N in (96, 256, 640, 1024, 1920, 2400, 3000, 3840, 4000, 4800, 5120, 6000, 7680, 8192, 9600, 10000, 15360, 16000, 16384, 17280, 32768):
            try:
                s = pa.Setup(N, tr, dt)
            except ValueError:
                continue
            if pa.kernel_name(s) in ("fourstep", "tiny"):
                s.close(); continue
            line = pa.describe(s).split("\n")[1]
            m = re.search(r"(?:threads|wg) (\d+).*?lds (\d+)", line)
            if not m:
                s.close(); continue
            th, lds = int(m.group(1)), int(m.group(2))
            occ = L.pffft_hip_route_occupancy(s.handle, 0, 1)
            by_lds, by_thr = (160 * 1024) // lds, 2048 // th
            print(f"{np.dtype(dt).name} {'cplx' if tr else 'real'} N={N:6d}: occupancy {occ}  (LDS admits {by_lds}, threads admit {by_thr})  {line.strip()[:110]}", flush=True)
            s.close()
