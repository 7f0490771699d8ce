import sys, os
sys.path.insert(0, os.getcwd())
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from sensorium_amd.argus_models import MouseModel
from sensorium_amd.synthetic import make_batch
dev = torch.device("cuda", 0)
params = bench.model_params(7); params["device"] = "cuda:0"; params["amp"] = True
torch.manual_seed(0)
model = MouseModel(params); model.set_ema(0.999)
batch = make_batch(32, 32, 36, 64, (7863,), seed=1, device=dev)
for _ in range(3): model.train_step(batch, sync_loss=False)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    model.train_step(batch, sync_loss=False)
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_stack_n=6)
rows = [e for e in ka if ("copy" in e.key.lower() or "clone" in e.key.lower() or "fill" in e.key.lower() or "zero" in e.key.lower()) ]
rows.sort(key=lambda e: -e.count)
for e in rows[:25]:
    print(e.count, e.key, "|", " <- ".join(s.split("/")[-1] for s in e.stack[:6]))
