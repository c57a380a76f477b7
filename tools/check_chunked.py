"""Single-GPU check of the N>1 frame-loop structure: processing the frames in K chunks
through TimeSeriesExchange must give the same series as one process() call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from upsp_processing_amd import engine, synthetic as syn, distributed as D
size, N, F = 256, 20000, 203
g = torch.Generator(device="cuda"); g.manual_seed(0)
pix = torch.randint(-1, size * size, (N,), generator=g, device="cuda", dtype=torch.int32)
frames = syn.synth_frames_torch(F, size, size)
pipe = engine.FramePipeline(1, size, size, N)
pipe.set_projection(0, pix)
ref = torch.empty((N, F), dtype=torch.float32, device="cuda")
pipe.process(frames, 0, rows_t=ref, want_rows=False)
s0 = [t.clone() for t in pipe.accumulators()]
pipe.reset()
shard = D.Shard(F, N, 0, 1)
ex = D.TimeSeriesExchange(shard, 4)
for k in range(4):
    c0, fc = ex.my_chunk(k)
    buf = torch.empty((N, fc), dtype=torch.float32, device="cuda")
    if fc:
        pipe.process(frames[c0:c0 + fc], first_frame=c0, rows_t=buf, want_rows=False)
    ex.submit(buf)
out = ex.finish()
assert torch.equal(out.view(torch.int32), ref.view(torch.int32))
s1 = pipe.accumulators()
ok = ~torch.isnan(s0[0])
assert torch.allclose(s0[0][ok], s1[0][ok], rtol=1e-12) and torch.allclose(s0[1][ok], s1[1][ok], rtol=1e-12)
# the same with only the visible nodes' rows travelling
pipe.reset()
ex2 = D.TimeSeriesExchange(shard, 4)
ex2.set_skipped(engine.skipped_nodes(pix, want_count=False)[0])
for k in range(4):
    c0, fc = ex2.my_chunk(k)
    buf = torch.empty((N, fc), dtype=torch.float32, device="cuda")
    if fc:
        pipe.process(frames[c0:c0 + fc], first_frame=c0, rows_t=buf, want_rows=False)
    ex2.submit(buf)
assert torch.equal(ex2.finish().view(torch.int32), ref.view(torch.int32))
# ... and with the gather writing the packed rows itself (row map)
pipe.reset()
ex3 = D.TimeSeriesExchange(shard, 4)
ex3.set_skipped(engine.skipped_nodes(pix, want_count=False)[0])
pipe.set_row_map(ex3.row_map())
for k in range(4):
    c0, fc = ex3.my_chunk(k)
    buf = torch.full((ex3.packed_rows(), fc), -7.0, dtype=torch.float32, device="cuda")
    if fc:
        pipe.process(frames[c0:c0 + fc], first_frame=c0, rows_t=buf, want_rows=False)
    ex3.submit(buf, packed=True)
assert torch.equal(ex3.finish().view(torch.int32), ref.view(torch.int32))
# ... and the same rows produced and exchanged as u16 (integer-valued series)
pipe.reset()
ex4 = D.TimeSeriesExchange(shard, 4)
ex4.set_skipped(engine.skipped_nodes(pix, want_count=False)[0])
for k in range(4):
    c0, fc = ex4.my_chunk(k)
    buf = torch.full((ex4.packed_rows(), fc), 9, dtype=torch.int32, device="cuda").to(torch.uint16)
    if fc:
        pipe.process(frames[c0:c0 + fc], first_frame=c0, rows_t=buf, want_rows=False)
    ex4.submit(buf, packed=True)
assert torch.equal(ex4.finish().view(torch.int32), ref.view(torch.int32))
pipe.set_row_map(None)
print("chunked == single-call: ok")
