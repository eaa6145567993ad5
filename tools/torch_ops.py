"""Which ATen ops (not ours) still launch kernels in one eager stage-3 step, and from where."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from torch.profiler import profile, ProfilerActivity

tr = bench.build_trainer(3, 16, 256, "cuda:0", graphs=False)
loader = iter(tr.train_loader)
for it in range(1, 3):
    tr.optimize_parameters(it, {**next(loader), "rate_ind": 2})
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    tr.optimize_parameters(3, {**next(loader), "rate_ind": 2})
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=False).table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=60))
print(prof.key_averages(group_by_stack_n=6).table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=50, max_src_column_width=110))
