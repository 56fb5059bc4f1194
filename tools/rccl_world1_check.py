"""The collectives bench.py issues at N > 1 (RCCL: barrier, MAX all-reduce of the region times, all_gather_object of the shard rows), run
with a world of ONE rank on one GPU - what a 1-GPU box can check of that path: the backend initialises with device_id, the calls return.
   python tools/rccl_world1_check.py"""
import datetime, os, time
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29531')
os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1'); os.environ.setdefault('LOCAL_RANK', '0')
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
t = time.perf_counter()
dist.init_process_group('nccl', device_id=torch.device('cuda', 0), timeout=datetime.timedelta(seconds=120))
dist.barrier()
x = torch.tensor([1.5, 2.5, 3.5], dtype=torch.float64, device='cuda:0')
dist.all_reduce(x, op=dist.ReduceOp.MAX)
rows = [None]
dist.all_gather_object(rows, (7, 0, 160, [0.1, 0.2], 1153.0, 4))
torch.cuda.synchronize()
dist.barrier()
dist.destroy_process_group()
print(f'rccl world-1 check ok in {time.perf_counter() - t:.1f} s: all_reduce {x.tolist()}, gathered {rows}')
