"""Does a second pass over a chunk come out of the 256 MiB Infinity Cache?  Two read passes (lo_hbm_stream_dev, read-only,
8 bytes per element over two arrays) over 2 x 2 GiB, chunk by chunk: pass, pass again, next chunk.  Reports the rate of
all bytes read against the chunk's working set, for plain and non-temporal loads."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linear_operator_amd import _hip
dev = torch.device("cuda", 0)
lib = _hip.load(); st = _hip.stream_ptr(dev)
n = 1 << 29
a = torch.empty(1024, dtype=torch.float32, device=dev)
b = torch.ones(n, dtype=torch.float32, device=dev); c = torch.ones(n, dtype=torch.float32, device=dev)
pb, pc = _hip.ptr(b), _hip.ptr(c)
import ctypes
def addr(p, off): return ctypes.c_void_p(p.value + 4 * off)
for nt in (0, 1):
    for passes in (1, 2, 3):
        for chunk_mb in (16, 32, 64, 96, 128, 192, 256, 512, 4096):
            m = chunk_mb * (1 << 20) // 8  # floats per array and chunk (working set = 8 m bytes)
            m = min(m, n)
            def sweep():
                for off in range(0, n, m):
                    for _ in range(passes):
                        lib.lo_hbm_stream_dev(2, 4, nt, _hip.ptr(a), addr(pb, off), addr(pc, off), 0.5, min(m, n - off), st)
            sweep(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); sweep(); e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            print(f"nt {nt} passes {passes} chunk {chunk_mb:5d} MB: {ms:8.3f} ms  {8.0 * n * passes / ms / 1e6:8.1f} GB/s "
                  f"({-(-n // m) * passes} launches)", flush=True)
