"""Row-pitch effect of pass B at a rank's full series length (configs[3]: F = 12 500 frames per rank, 100 000 per run):
upsp_rows_from_pixel_series writes [N][ld] f32 rows from a compact [active pixel][frames] u16 buffer; ld = F (the reference's
tight [N][F] layout, 50 000-byte rows) against ld rounded up to 64 floats (rows on 256-byte boundaries).
   python tools/pitch_probe.py [F]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from upsp_processing_amd import _capi
F = int(sys.argv[1]) if len(sys.argv) > 1 else 12500
N, A = 500958, 66000
cp = (F + 63) // 64 * 64
g = torch.Generator(device="cuda"); g.manual_seed(1)
compact = torch.randint(0, 4000, (A, cp), generator=g, device="cuda", dtype=torch.int32).to(torch.uint16)
node_k = torch.randint(0, A, (N,), generator=g, device="cuda", dtype=torch.int32)
node_k[torch.rand(N, generator=g, device="cuda") < 0.6] = -1        # 60 % of the rows hold no data (constant fill)
skipped = (node_k < 0).to(torch.uint8)
s = torch.zeros(N, dtype=torch.float64, device="cuda"); ss = torch.zeros_like(s)
L = _capi.lib()
for ld in (F, cp, cp + 64):
    rows = torch.empty((N, ld), dtype=torch.float32, device="cuda")
    def run():
        _capi.check(L.upsp_rows_from_pixel_series(C.c_void_p(compact.data_ptr()), cp, C.c_void_p(node_k.data_ptr()), C.c_void_p(skipped.data_ptr()),
                                                  N, F, C.c_void_p(rows.data_ptr()), ld, C.c_void_p(s.data_ptr()), C.c_void_p(ss.data_ptr()),
                                                  C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print("F = %d  row pitch %6d floats (%7d B, %s)  pass B %.3f ms  = %.2f TB/s of row bytes" % (
        F, ld, ld * 4, "256-B aligned" if ld % 64 == 0 else "unaligned", ms, N * F * 4 / ms / 1e9), flush=True)
    del rows
