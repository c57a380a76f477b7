"""One registration sub-batch (64 frames of 1 Mpix) for rocprofv3 --pmc passes of ecc_sums_kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from upsp_processing_amd import engine, synthetic as syn
size, F, N = 1024, 64, 100000
frames = torch.empty((F, size, size), dtype=torch.uint16, device="cuda")
syn.synth_frames_torch(F, size, size, out=frames)
pix = torch.randint(0, size * size, (N,), device="cuda", dtype=torch.int32)
pipe = engine.FramePipeline(1, size, size, N, registration=1)
pipe.set_reference(0, frames[0].to(torch.float32))
pipe.set_projection(0, pix)
for _ in range(2):
    pipe.reset()
    pipe.process(frames, 0, want_rows=True)
torch.cuda.synchronize()
print(pipe.ecc_stats())
