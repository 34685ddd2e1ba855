"""One 1080p/1024-spp frame as S sample-slices traced concurrently (S contexts with
sample_offset = s, sample_stride = S, on streams that overlap), against one launch."""
import sys, time
import torch
sys.path.insert(0, ".")
import pathtracer_amd as pa
from pathtracer_amd.distributed import stream_on_another_queue

first = torch.cuda.Stream()
torch.cuda.set_stream(first)
streams = [first]
keep = []
while len(streams) < 4:
    s, spare = stream_on_another_queue(first)
    keep += spare
    streams.append(s)
P = pa.Params(1920, 1080, 500.0, 1024, 8)
for S in (1, 2, 4):
    rs = [pa.Renderer(P, pa.objects, seed=0, sample_offset=s, sample_stride=S) for s in range(S)]
    for r, st in zip(rs, streams):
        r.set_stream(st.cuda_stream)
    for rep in range(2):
        for r in rs:
            r.reset(); r.trace(32)
    torch.cuda.synchronize()
    ts = []
    for rep in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for r in rs:
            r.reset(); r.trace(1024 // S)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f"S={S}: {min(ts):.3f} ms best, {sum(ts)/len(ts):.3f} mean; segments {sum(r.segments for r in rs)}")
    for r in rs:
        r.close() if hasattr(r, "close") else None
