#!/usr/bin/env python3
"""Step time of the metric configuration with y1 left unmaterialised on a chosen set of blocks (dwn_block_args.y1_mode per block):
python3 tools/y1_block_ab.py 0123 01234 0123456 ..."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch

from bench import model_params
from sensorium_amd.argus_models import MouseModel
from sensorium_amd.dwiseneuro import InvertedResidual3d
from sensorium_amd.synthetic import make_batch


def main():
    dev = torch.device("cuda:0")
    p = model_params(7)
    p["device"] = str(dev); p["amp"] = True
    model = MouseModel(p)
    model.set_ema(0.999)
    batch = make_batch(32, 32, 36, 64, (7863,), device=dev)
    blocks = [m for m in model.nn_module.modules() if isinstance(m, InvertedResidual3d)]
    for spec in sys.argv[1:]:
        for i, b in enumerate(blocks):
            b._dwn_y1_mode = 2 if str(i) in spec else 1
        for _ in range(4):
            model.train_step(batch, sync_loss=False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            model.train_step(batch, sync_loss=False)
        e1.record(); e1.synchronize()
        print(f"y1-free blocks {spec:10s}: {e0.elapsed_time(e1) / 20:.3f} ms/step", flush=True)


if __name__ == "__main__":
    main()
