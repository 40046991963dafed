"""Lab: where a 64 MiB call's time outside the stream goes (cfg 2, engine 5): the Python method, the bare ctypes call with
arguments built once, with and without the library's profiling events.  Wall clock per call over 200 calls."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from aha_amd import AC, synth
from aha_amd import _native as N
from aha_amd.ac import _params

blob, offs, nf = synth.keys(2)
ac = AC.compile_packed(blob, offs)
corpus, doc = synth.corpus(2, blob, offs, nf, n_bytes=64 << 20)
dc = torch.from_numpy(corpus).cuda()
dd = torch.from_numpy(doc.astype(np.int64)).cuda()
dho = torch.zeros(doc.size, dtype=torch.int64, device="cuda")
out = torch.zeros((400000, 3), dtype=torch.int32, device="cuda")
for prof in (True, False):
    ac.set_profiling(prof)
    for _ in range(10):
        ac.match_batch_device(dc, dd, out, dho)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        ac.match_batch_device(dc, dd, out, dho)
    t_py = (time.perf_counter() - t0) / 200
    p = _params(False, None)
    n = C.c_uint64(0)
    s = C.c_void_p(torch.cuda.current_stream(dc.device).cuda_stream)
    f = N.lib().aha_ac_match_batch_device
    args = (ac._h, dc.data_ptr(), dd.data_ptr(), doc.size - 1, dc.numel(), C.byref(p), out.data_ptr(), out.numel() // 3, dho.data_ptr(), C.byref(n), s)
    for _ in range(10):
        f(*args)
    t0 = time.perf_counter()
    for _ in range(200):
        f(*args)
    t_c = (time.perf_counter() - t0) / 200
    tm = ac.last_timing() if prof else None
    print(f"profiling {prof}: python method {t_py * 1e6:.1f} us per call, bare ctypes call {t_c * 1e6:.1f} us" +
          (f", stream time {tm['ms_total'] * 1e3:.1f} us" if tm else ""), flush=True)
