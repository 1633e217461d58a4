"""What the memory system of this GPU delivers to plain streaming kernels (torch's own copy / fill / reduce kernels), for comparison with the
voxel update's measured HBM traffic per second (profiles/r03_pmc_tsdf_update.json x blocks / launch time): copy = 1 read : 1 write, the
voxel update 6.1 KB read : 4.6 KB written per block.  Sizes well beyond the 256 MB memory-side cache."""
import json
import sys
import torch

def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps

out = {}
for mb in (128, 512, 2048):
    n = mb * (1 << 20) // 4
    a = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
    b = torch.empty_like(a)
    t_copy = timed(lambda: b.copy_(a))
    t_fill = timed(lambda: b.fill_(1.0))
    t_read = timed(lambda: a.sum())
    out["%d MB" % mb] = {"copy_TBps_read_plus_write": round(2 * mb * (1 << 20) / t_copy / 1e12, 3), "fill_TBps": round(mb * (1 << 20) / t_fill / 1e12, 3),
                         "reduce_TBps": round(mb * (1 << 20) / t_read / 1e12, 3)}
    del a, b
print(json.dumps(out))
